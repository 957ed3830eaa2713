#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in the BUILT library, read from the code objects' metadata (no
compiler run, no GPU):    python tools/kernel_table.py [filter]
The .hip_fatbin section holds one clang offload bundle per translation unit; each gfx950 entry is an ELF whose
NT_AMDGPU_METADATA note lists .name / .vgpr_count / .agpr_count / .sgpr_count / .private_segment_fixed_size (scratch
bytes per lane) / .group_segment_fixed_size (static LDS) per kernel.  tests/test_kernel_resources.py holds the hot
kernels to "no scratch" with it."""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path):
    """gfx950 ELF images inside the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
        blob = open(fat, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), blob):
        base = m.start()
        n, = struct.unpack_from("<Q", blob, base + 24)
        pos = base + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[base + off:base + off + size])
    return out


def kernels(lib_path=None):
    lib_path = lib_path or os.path.join(ROOT, "hydrainfer_amd", "lib", "libhydra_hip.so")
    rows = []
    for img in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img); f.flush()
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'")
            # a kernel's keys come in alphabetical order: .agpr_count first, .wavefront_size last
            if k in ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count",
                     "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size", "kernarg_segment_size", "wavefront_size"):
                if cur is None:
                    cur = {}
                cur[k] = int(v)
                if k == "wavefront_size":
                    if "symbol" in cur:
                        rows.append(cur)
                    cur = None
            elif k == "symbol" and v.endswith(".kd"):
                if cur is None:
                    cur = {}
                cur["symbol"] = v[:-3]
    names = subprocess.run(["c++filt"], input="\n".join(r["symbol"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["name"] = n.replace("(anonymous namespace)::", "").replace("hx::", "").replace("void ", "").split("(")[0]
    return rows


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    for r in sorted(kernels(), key=lambda r: r["name"]):
        if flt in r["name"]:
            print(f"{r['name'][:84]:84s} vgpr {r.get('vgpr_count', 0):4d} agpr {r.get('agpr_count', 0):4d} sgpr {r.get('sgpr_count', 0):4d} "
                  f"scratch {r.get('private_segment_fixed_size', 0):5d} lds {r.get('group_segment_fixed_size', 0):6d} spill {r.get('vgpr_spill_count', 0)}")
