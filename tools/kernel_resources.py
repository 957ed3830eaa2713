#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_resources.py hydrainfer_amd/csrc/gemm_xreg.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = (["-fno-slp-vectorize"] if "attn_fwd" in src else []) + ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
for c in rows:
    name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("hx::", "").split("(")[0].replace("void ", "")
    if flt and flt not in name:
        continue
    print(f"{name:70s} VGPR {c.get('VGPRs','?'):>4} AGPR {c.get('AGPRs','?'):>4} spill {c.get('VGPRs Spill','?'):>3} scratch {c.get('ScratchSize [bytes/lane]','?'):>4} "
          f"occ {c.get('Occupancy [waves/SIMD]','?')} LDS {c.get('LDS Size [bytes/block]','?')}")
