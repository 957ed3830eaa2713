"""CPU: the C-ABI shared library loads and exports every symbol include/hydra_hip.h
declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

from hydrainfer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(experimental=False):
    """hx_* functions declared by include/hydra_hip.h (the product ABI) or, experimental=True, by
    include/hydra_hip_experimental.h (exported only by `make EXPERIMENTS=1` builds)."""
    names = []
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not hdr.endswith(".h") or hdr.endswith("_experimental.h") != experimental:
            continue
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(hx_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    declared = _declared()
    assert len(declared) >= 19
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared if not hasattr(handle, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"


def test_experiments_are_not_in_the_default_library():
    """Rejected experiments and microbenchmarks (four-heads decode attention, stream probes) ship
    only in `make EXPERIMENTS=1` builds: either all of hydra_hip_experimental.h is exported or none of it, and the
    ctypes binding types exactly those names."""
    exp = _declared(experimental=True)
    assert exp == sorted(_lib._EXPERIMENTAL_SIGNATURES) and len(exp) >= 2
    handle = ctypes.CDLL(_lib.LIB_PATH)
    present = [n for n in exp if hasattr(handle, n)]
    assert present in ([], exp), present
    assert _lib.has_experiments() == bool(present)
    assert not set(exp) & set(_declared())


def test_python_binding_covers_the_header():
    assert sorted(_lib.exported_symbols()) == _declared()
    lib = _lib.lib()
    assert lib.hx_abi_version() == _lib.HX_ABI_VERSION == 3
    assert lib.hx_strerror(0) == b"ok"
    assert b"data type" in lib.hx_strerror(-1)


def test_attn_args_struct_layout_matches_header():
    # field order/size of hx_attn_args as declared in the header
    src = open(os.path.join(ROOT, "include", "hydra_hip.h")).read()
    body = re.search(r"typedef struct hx_attn_args \{(.*?)\} hx_attn_args;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split()[-1] if "," not in decl else None
        if names is None:
            first, *rest = decl.split(",")
            fields.append(first.split()[-1].lstrip("*"))
            fields += [r.strip().lstrip("*") for r in rest]
        else:
            fields.append(names.lstrip("*"))
    assert fields == [f[0] for f in _lib.hx_attn_args._fields_]


def test_cpu_tensors_are_refused_not_emulated():
    import pytest
    import torch
    from hydrainfer_amd._C.kernel.norm import rms_norm
    x = torch.randn(2, 8)
    with pytest.raises(_lib.HydraHipError):
        rms_norm(torch.empty_like(x), x, torch.ones(8), 1e-5)


def test_op_modules_match_the_reference_stubs():
    """Module paths, function names and positional parameter lists of hydrainfer._C.* as the
    reference's own .pyi stubs declare them (tests/golden/op_signatures.json, extracted by
    tests/golden/generate_stub_signatures.py).  flash_infer is the one package deliberately absent
    (north_star: the flashinfer path is removed)."""
    import importlib
    import inspect
    import json
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "op_signatures.json")))
    missing = []
    for module, fns in want.items():
        if module.endswith("flash_infer"):
            continue
        mod = importlib.import_module(f"hydrainfer_amd.{module}")
        for name, params in fns.items():
            fn = getattr(mod, name, None)
            if fn is None:
                missing.append(f"{module}.{name}")
                continue
            got = [p.name for p in inspect.signature(fn).parameters.values()
                   if p.default is inspect.Parameter.empty and p.kind == p.POSITIONAL_OR_KEYWORD]
            assert len(got) == len(params), f"{module}.{name}: {got} vs reference {params}"
    assert not missing, missing


def test_xreg_split_plan_is_stable():
    """The packed layout of hx_pack_decode_weight_xreg depends on the split plan for (N, K): weights packed by one
    build must be readable by the next.  Pins the plan for the LLaVA-1.5 7B / 13B projections and two small shapes
    (256 CUs assumed when no device is present; an MI355X has 256)."""
    l = _lib.lib()
    want = {(12288, 4096): 1, (22016, 4096): 1, (4096, 11008): 4, (4096, 4096): 1,
            (15360, 5120): 1, (27648, 5120): 1, (5120, 13824): 4, (48, 64): 1, (64, 2816): 1}
    for (n, k), s in want.items():
        assert l.hx_linear_decode_xreg_splits(n, k) == s, (n, k)
        assert l.hx_linear_decode_xreg_supported(32, n, k) == 1
        assert l.hx_linear_decode_xreg_workspace_bytes(32, n, k) == s * 32 * n * 4
    # 33 .. 64 rows: the wide kernel reads the SAME packing with half the k-steps per wave (twice the slabs) where the
    # packing's k-steps per wave halve to a built count: all 7B projections and, since round 5, the 13B ones
    # (40 -> 20 with one row group per unit; 27 -> 14 + 13)
    for (n, k), s in {(12288, 4096): 2, (4096, 11008): 8, (4096, 4096): 2, (64, 2816): 2,
                      (15360, 5120): 2, (5120, 13824): 8}.items():
        assert l.hx_linear_decode_xreg_supported(64, n, k) == 1 and l.hx_linear_decode_xreg_supported(33, n, k) == 1
        assert l.hx_linear_decode_xreg_workspace_bytes(64, n, k) == s * 64 * n * 4, (n, k)
    assert l.hx_gate_up_xreg_supported(64, 11008, 4096, 1) == 1 and l.hx_gate_up_xreg_workspace_bytes(64, 11008, 4096) == 2 * 64 * 22016 * 4
    assert l.hx_gate_up_xreg_supported(64, 13824, 5120, 1) == 1 and l.hx_gate_up_xreg_workspace_bytes(64, 13824, 5120) == 2 * 64 * 27648 * 4
    assert l.hx_norm_xreg_supported(64, 15360, 5120, 0) == 1 and l.hx_norm_xreg_supported(64, 5120, 13824, 0) == 0
    assert l.hx_linear_decode_xreg_supported(64, 1024, 3584) == 0       # 29 k-steps per wave: no built half
    assert l.hx_linear_decode_xreg_supported(65, 4096, 4096) == 0
    assert l.hx_gate_up_silu_xreg_supported(33, 11008, 4096) == 0      # the fused silu*mul epilogue: <= 32 rows
    assert l.hx_norm_xreg_supported(64, 12288, 4096, 0) == 1 and l.hx_norm_xreg_supported(64, 4096, 11008, 0) == 0
    assert l.hx_gate_up_silu_xreg_supported(32, 11008, 4096) == 1 and l.hx_gate_up_silu_xreg_supported(32, 13824, 5120) == 1
    assert l.hx_norm_xreg_supported(32, 12288, 4096, 0) == 1 and l.hx_norm_xreg_supported(32, 4096, 11008, 0) == 0
    assert l.hx_fragment_major_elems(5, 4096) == 16 * 4096 and l.hx_fragment_major_elems(32, 64) == 32 * 64
