"""CPU: what the compiler made of the hot kernels, read from the BUILT library's code-object metadata
(tools/kernel_table.py; no GPU, no compiler run).  A compiler or flag change that makes a decode-path kernel spill is
a silent 5-40 % loss (round 5: 45 spilled registers in the 40-k-step norm-fused GEMM cost the 13B step 3.6 %; the
round-4 head_dim-256 decode attention ran 67 us with scratch, 46 without) — this test makes it loud."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_table  # noqa: E402


@pytest.fixture(scope="module")
def table():
    rows = kernel_table.kernels()
    assert len(rows) > 300, "the library's code objects were not found (layout of .hip_fatbin changed?)"
    return rows


HOT = ("attn_decode_kernel<", "attn_decode_gqa_kernel<", "attn_decode_combine_kernel<", "gemm_xreg_kernel<",
       "gemm_xreg_wide_kernel<", "gemm_packed_kernel<", "decode_step_head_kernel<", "add_rms_norm_slab_kernel<",
       "argmax_rows_kernel<", "silu_mul_slab_kernel<", "set_kv_cache", "rope_cache", "gather_copy_blocks",
       "copy_words2_kernel", "collect_errors_kernel", "attn_fwd32_kernel<", "attn_fwd32p_kernel<")


def test_decode_path_kernels_use_no_scratch(table):
    hot = [r for r in table if r["name"].startswith(HOT)]
    assert len(hot) > 150
    bad = [(r["name"], r.get("private_segment_fixed_size"), r.get("vgpr_spill_count")) for r in hot
           if r.get("private_segment_fixed_size", 0) != 0 or r.get("vgpr_spill_count", 0) != 0]
    assert not bad, f"kernels of the decode path with scratch / spilled registers: {bad[:8]}"
    for r in hot:
        assert r.get("wavefront_size") == 64, r["name"]


def test_register_budgets_of_the_benchmarked_instantiations(table):
    by = {r["name"]: r for r in table}
    # fused decode attention, head_dim 128: <= 128 registers = 4 workgroups of 4 waves per CU (the 1024 workgroups of a
    # batch-32 launch are all resident at once)
    for dt in ("BF16", "F16"):
        for ranked in ("false", "true"):      # the static grid, and the RANKED form the benchmark's big batches run
            r = by[f"attn_decode_kernel<{dt}, 128, 4, true, true, {ranked}>"]
            assert r["vgpr_count"] <= 128, r                            # (.vgpr_count is the unified total, AGPRs included)
            assert r["group_segment_fixed_size"] <= 40 * 1024           # four per CU inside 160 KiB
            assert r["kernarg_segment_size"] >= 64                      # 7 pointers + 2 ints in front of the struct: all preloaded
    # activations-in-registers GEMMs: one wave per SIMD by design (x and a weight buffer set fill the 512 registers)
    for name in ("gemm_xreg_kernel<BF16, 2, 32, 1, 0, 1, 0>", "gemm_xreg_kernel<BF16, 2, 32, 0, 0, 1, 0>",
                 "gemm_xreg_kernel<BF16, 2, 22, 0, 0, 0, 0>", "gemm_xreg_kernel<BF16, 2, 40, 1, 0, 1, 0>",
                 "gemm_xreg_kernel<BF16, 2, 40, 0, 0, 1, 0>", "gemm_xreg_kernel<BF16, 2, 27, 0, 0, 0, 0>",
                 "gemm_xreg_wide_kernel<BF16, 16, 1, 2, 0>", "gemm_xreg_wide_kernel<BF16, 16, 1, 2, 1>",
                 "gemm_xreg_wide_kernel<BF16, 11, 0, 2, 0>", "gemm_xreg_wide_kernel<BF16, 20, 1, 1, 0>",
                 "gemm_xreg_wide_kernel<BF16, 14, 0, 2, 0>"):
        r = by[name]
        assert r["vgpr_count"] <= 512 and r.get("private_segment_fixed_size", 0) == 0, r


def test_kernel_arguments_of_the_hot_kernels_fit_the_preload_window(table):
    """The kernels whose first loads hang on their arguments take them as leading scalars (gemm_xreg.hip, KERNARG
    PRELOADING): the struct that follows must not have displaced them — the explicit arguments in front of the struct
    are 9 x 8-byte / 4-byte slots = 56 bytes (14 dwords, what gfx950 preloads)."""
    by = {r["name"]: r for r in table}
    r = by["gemm_xreg_kernel<BF16, 2, 32, 0, 0, 1, 0>"]
    assert r["kernarg_segment_size"] >= 56 + 100          # 5 pointers + 4 ints, then the by-value struct


def test_prefill_attention_has_no_flat_memory_instructions():
    """attn_fwd.hip's persistent kernel waits with a COUNTED s_waitcnt vmcnt(N) at the seam between two items (first tile
    of the next item in flight behind the O stores of this one).  That is sound because a wave's vector-memory operations
    complete in issue order — except flat_* ones (MI355X_MICROARCH.md; GFX9 ISA, S_WAITCNT).  The built prefill kernels
    must therefore contain none."""
    import re
    import subprocess
    import tempfile
    lib = os.path.join(ROOT, "hydrainfer_amd", "lib", "libhydra_hip.so")
    n_fwd = n_flat = 0
    for img in kernel_table.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img); f.flush()
            syms = subprocess.run([os.path.join(kernel_table.LLVM, "llvm-readelf"), "--symbols", "--wide", f.name],
                                  capture_output=True, text=True).stdout
            if "attn_fwd32" not in syms:
                continue
            asm = subprocess.run([os.path.join(kernel_table.LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name],
                                 capture_output=True, text=True).stdout
        cur = None
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                n_fwd += "attn_fwd32" in cur
            elif cur and "attn_fwd32" in cur and re.search(r"\bflat_(load|store|atomic)", line):
                n_flat += 1
    assert n_fwd >= 8, "the prefill kernels were not found in the library's code objects"
    assert n_flat == 0, f"{n_flat} flat_* memory instructions in the prefill attention kernels: the counted vmcnt seam is unsound"
