"""-m gpu: the one-launch decode chain (csrc/decode_chain.hip) against the eight-launch sequence
of ops it replaces — o GEMM, add+rms_norm, gate|up GEMM, silu*mul, down GEMM, add+rms_norm, next
layer's qkv GEMM (hydrainfer/model/model_forward.py:84-105,72-77).  Those ops are themselves held
to the oracle / the reference's goldens in test_gpu_ops.py; the chain must be BIT-identical to
them, on every call of a long back-to-back run over the same buffers (a stale hand-over inside the
launch would show up as a mismatch on a warm cache, not on the first call)."""
import pytest
import torch

from hydrainfer_amd._C.kernel import gemm as gemm_mod

from hydrainfer_amd import _lib

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (_lib.has_experiments() if __import__("os").path.exists(_lib.LIB_PATH) else False),
                                 reason="decode_chain.hip is only in `make EXPERIMENTS=1` builds of libhydra_hip.so")]
DEV = "cuda:0"


def _mk(shape, g, dt, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dt).to(DEV)


def _separate(attn_out, h_in, w_o, w_gu, w_dn, w_qkv, n_post, n_next, eps):
    """The launch sequence of LlamaForCausalLM._decode_hidden_hip_gemm for one layer."""
    from hydrainfer_amd._C.kernel import activation, gemm, norm
    M, hid = h_in.shape
    inter = w_dn.shape[1]
    ws = torch.empty(max(gemm.workspace_floats(M, 2 * inter, hid), gemm.workspace_floats(M, hid, inter),
                         gemm.workspace_floats(M, hid, attn_out.shape[1]),
                         gemm.workspace_floats(M, w_qkv.shape[0], hid) if w_qkv is not None else 0),
                     dtype=torch.float32, device=DEV)
    h = h_in.clone()
    x_post = torch.empty_like(h)
    s = gemm.linear_decode_partial(attn_out, w_o, ws)
    norm.add_rms_norm_slabs(x_post, h, ws, s, n_post, eps)
    h_mid = h.clone()
    s = gemm.linear_decode_partial(x_post, w_gu, ws)
    act = activation.silu_and_mul_slabs(ws, s, M, inter, h.dtype)
    s = gemm.linear_decode_partial(act, w_dn, ws)
    x_next = torch.empty_like(h)
    norm.add_rms_norm_slabs(x_next, h, ws, s, n_next, eps)
    out = dict(h_mid=h_mid, x_post=x_post, act=act, h_out=h, x_next=x_next)
    if w_qkv is not None:
        s = gemm.linear_decode_partial(x_next, w_qkv, ws)
        out["qkv"] = ws[: s * M * w_qkv.shape[0]].view(s, M, w_qkv.shape[0]).clone()
    return out


_PACK_CACHE = {}


def _packed(w):
    key = (w.data_ptr(), tuple(w.shape))
    if key not in _PACK_CACHE:
        if len(_PACK_CACHE) > 16:
            _PACK_CACHE.clear()
        _PACK_CACHE[key] = (w, gemm_mod.pack_weight(w))
    return _PACK_CACHE[key][1]


def _chain(attn_out, h_in, w_o, w_gu, w_dn, w_qkv, n_post, n_next, eps, bufs=None):
    from hydrainfer_amd._C.kernel import gemm
    M, hid = h_in.shape
    inter, q_size = w_dn.shape[1], attn_out.shape[1]
    dt = h_in.dtype
    if bufs is None:
        bufs = dict(h_mid=torch.empty_like(h_in), h_out=torch.empty_like(h_in), x_post=torch.empty_like(h_in),
                    x_next=torch.empty_like(h_in), act=torch.empty((M, inter), dtype=dt, device=DEV),
                    ws=torch.empty(gemm.chain_workspace_floats(M, hid, inter, q_size), dtype=torch.float32, device=DEV),
                    qkv=(torch.empty(gemm.workspace_floats(M, w_qkv.shape[0], hid), dtype=torch.float32, device=DEV)
                         if w_qkv is not None else None),
                    sync=torch.zeros(gemm.SYNC_WORDS, dtype=torch.int32, device=DEV))
    bufs["sync"].zero_()
    pk = lambda w: None if w is None else _packed(w)
    s = gemm.decode_chain(attn_out, h_in, pk(w_o), pk(w_gu), pk(w_dn), pk(w_qkv), inter, n_post, n_next, eps,
                          bufs["h_mid"], bufs["h_out"],
                          bufs["x_post"], bufs["act"], bufs["x_next"], bufs["qkv"], bufs["ws"], bufs["sync"])
    out = {k: bufs[k] for k in ("h_mid", "x_post", "act", "h_out", "x_next")}
    if w_qkv is not None:
        out["qkv"] = bufs["qkv"][: s * M * w_qkv.shape[0]].view(s, M, w_qkv.shape[0])
    return out, bufs


def _weights(hid, inter, q_size, qkv_n, dt, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(w_o=_mk((hid, q_size), g, dt, 0.02), w_gu=_mk((2 * inter, hid), g, dt, 0.02),
                w_dn=_mk((hid, inter), g, dt, 0.02), w_qkv=_mk((qkv_n, hid), g, dt, 0.02) if qkv_n else None,
                n_post=(1 + 0.1 * torch.randn(hid, generator=g)).to(dt).to(DEV),
                n_next=(1 + 0.1 * torch.randn(hid, generator=g)).to(dt).to(DEV))


def _check(got, want, what):
    from hydrainfer_amd._C.kernel import gemm
    for k in want:
        assert torch.equal(got[k], want[k]), f"{what}: {k} differs ({(got[k] != want[k]).sum().item()} elements)"


SHAPES = [  # hidden, inter, q_size, qkv_n
    (512, 1024, 512, 1536),      # tiny model of the engine tests
    (4096, 11008, 4096, 12288),  # LLaVA-1.5-7B
    (5120, 13824, 5120, 15360),  # LLaVA-1.5-13B
    (1024, 2816, 1024, 0),       # last layer: no next qkv; inter with a short last K split
]


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 7, 16, 32])
@pytest.mark.parametrize("shape", SHAPES)
def test_chain_equals_separate_launches(shape, M, dt):
    hid, inter, q_size, qkv_n = shape
    if M not in (7, 32) and hid > 4096:
        pytest.skip("13B shape: two batch sizes are enough")
    w = _weights(hid, inter, q_size, qkv_n, dt, seed=hid + M)
    g = torch.Generator().manual_seed(M)
    attn_out, h_in = _mk((M, q_size), g, dt), _mk((M, hid), g, dt)
    want = _separate(attn_out, h_in, eps=1e-5, **w)
    got, bufs = _chain(attn_out, h_in, eps=1e-5, **w)
    torch.cuda.synchronize()
    assert int(bufs["sync"][gemm_mod.SYNC_ERR]) == 0, "a dependency wait timed out"
    _check(got, want, f"{shape} M={M} {dt}")


@pytest.mark.parametrize("r", ["1,1,1,1", "2,4,2,2", "4,2,4,1"])
def test_chain_item_sizes_do_not_change_results(r):
    """row groups per wave (the work-item size of each GEMM phase) is a tuning knob only."""
    from hydrainfer_amd import _lib
    dt, M = torch.bfloat16, 32
    hid, inter, q_size, qkv_n = 4096, 11008, 4096, 12288
    w = _weights(hid, inter, q_size, qkv_n, dt, seed=11)
    g = torch.Generator().manual_seed(2)
    attn_out, h_in = _mk((M, q_size), g, dt), _mk((M, hid), g, dt)
    want = _separate(attn_out, h_in, eps=1e-5, **w)
    names = ("chain_r_o", "chain_r_gu", "chain_r_down", "chain_r_qkv")
    try:
        for n, v in zip(names, r.split(",")):
            assert _lib.lib().hx_debug_set_option(n.encode(), int(v)) == 0
        got, bufs = _chain(attn_out, h_in, eps=1e-5, **w)
        torch.cuda.synchronize()
        assert int(bufs["sync"][gemm_mod.SYNC_ERR]) == 0
        _check(got, want, f"R={r}")
    finally:
        for n, v in zip(names, (1, 2, 2, 1)):
            _lib.lib().hx_debug_set_option(n.encode(), v)


def test_chain_back_to_back_on_warm_buffers():
    """200 launches over the SAME scratch buffers with changing inputs, interleaved with a kernel
    that keeps other lines of the caches busy: every output of every launch equals the separate
    launches.  This is the test a stale hand-over (a consumer reading an old copy of a line
    another workgroup rewrote) cannot pass."""
    dt, M = torch.bfloat16, 32
    hid, inter, q_size, qkv_n = 4096, 11008, 4096, 12288
    w = _weights(hid, inter, q_size, qkv_n, dt, seed=5)
    g = torch.Generator().manual_seed(9)
    inputs = [(_mk((M, q_size), g, dt), _mk((M, hid), g, dt)) for _ in range(8)]
    wants = [_separate(a, h, eps=1e-5, **w) for a, h in inputs]
    bufs = None
    noise = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for it in range(200):
        a, h = inputs[it % 8]
        if it % 3 == 0:
            noise.add_(1)
        got, bufs = _chain(a, h, eps=1e-5, bufs=bufs, **w)
        if it % 8 == 7 or it < 8:
            torch.cuda.synchronize()
            assert int(bufs["sync"][gemm_mod.SYNC_ERR]) == 0
            _check(got, wants[it % 8], f"iteration {it}")


def test_chain_rejects_what_it_cannot_run():
    from hydrainfer_amd import _lib
    from hydrainfer_amd._C.kernel import gemm
    assert not gemm.chain_supported(33, 4096, 11008, 4096, torch.bfloat16)       # more than 32 rows
    assert not gemm.chain_supported(8, 4000, 11008, 4096, torch.bfloat16)        # hidden % 256
    assert not gemm.chain_supported(8, 4096, 11008, 4096, torch.float32)
    dt, M, hid, inter = torch.bfloat16, 4, 512, 1024
    w = _weights(hid, inter, hid, 0, dt, seed=1)
    g = torch.Generator().manual_seed(1)
    attn_out, h_in = _mk((M, hid), g, dt), _mk((M, hid), g, dt)
    ws = torch.empty(gemm.chain_workspace_floats(M, hid, inter, hid), dtype=torch.float32, device=DEV)
    sync = torch.zeros(gemm.SYNC_WORDS, dtype=torch.int32, device=DEV)
    e = torch.empty_like(h_in)
    act = torch.empty((M, inter), dtype=dt, device=DEV)
    with pytest.raises(_lib.HydraHipError):   # h_mid aliases h_in: a buffer would be written twice
        gemm.decode_chain(attn_out, h_in, _packed(w["w_o"]), _packed(w["w_gu"]), _packed(w["w_dn"]), None, inter,
                          w["n_post"], w["n_next"], 1e-5, h_in, e, torch.empty_like(e), act, torch.empty_like(e), None, ws, sync)
    with pytest.raises(_lib.HydraHipError):   # workspace too small
        gemm.decode_chain(attn_out, h_in, _packed(w["w_o"]), _packed(w["w_gu"]), _packed(w["w_dn"]), None, inter,
                          w["n_post"], w["n_next"], 1e-5, torch.empty_like(e), e, torch.empty_like(e), act, torch.empty_like(e), None, ws[:100], sync)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_model_decode_chain_equals_eight_launch_path(dt):
    """A 3-layer 7B-width model: 6 decode steps with the chain == the same steps with separate
    launches (hidden state bit-identical, KV pool bit-identical)."""
    from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
    sh = LlamaShape(1024, 2816, 3, 8, 8, 128, 2048)
    outs = []
    for chain in (False, True):
        model = LlamaForCausalLM.random_init(sh, dt, DEV, seed=3)
        model.use_chain = chain
        model.use_xreg = False   # the chain reproduces the LDS-slice GEMM's accumulation order
        r = DecodeRunner(model, RunnerConfig(batch=5, prompt_len=40, n_generate=8, use_graph=False), seed=4)
        g = torch.Generator().manual_seed(0)
        r.prefill(torch.randint(5, 2000, (5, 40), generator=g).to(DEV))
        for _ in range(6):
            r.step()
        torch.cuda.synchronize()
        if chain:
            assert int(model.chain_sync[:, gemm_mod.SYNC_ERR].abs().sum()) == 0
        outs.append((r.generated(), r.pool.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
