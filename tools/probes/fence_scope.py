#!/usr/bin/env python3
"""Which acquire / release fence scopes do the decode step's dispatches carry?  (round-4 review item 1b: a system-scope
release per launch boundary would mean an L2 write-back 164 times per step — a hypothesis to test before building on it.)
Decision it serves: whether launching the plan's kernels through another API (hipExtLaunchKernel flags / a different queue
setup) could shorten the 1.65 us per launch boundary that tools/null_layer.py measures.
Runs a tiny-model decode step under AMD_LOG_LEVEL=4 three ways (eager launches, launch plan, hipGraph) in a child process
each, and counts the 'Dispatch Header' forms ROCclr logs per mode."""
import collections, os, re, subprocess, sys

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
mode = sys.argv[1]
dev = torch.device("cuda:0")
sh = LlamaShape(1024, 2816, 2, 8, 8, 128, 2048)
model = LlamaForCausalLM.random_init(sh, torch.bfloat16, dev, seed=3)
r = DecodeRunner(model, RunnerConfig(batch=8, prompt_len=40, n_generate=16, use_graph=mode != "eager", executor="plan" if mode == "plan" else "graph"), seed=4)
g = torch.Generator().manual_seed(0)
r.prefill(torch.randint(5, 2000, (8, 40), generator=g).to(dev))
for _ in range(3):
    r.step()
torch.cuda.synchronize()
print("MARK-BEGIN", file=sys.stderr, flush=True)
for _ in range(2):
    r.step()
torch.cuda.synchronize()
print("MARK-END", file=sys.stderr, flush=True)
'''

out = []
for mode in ("eager", "plan", "graph"):
    env = dict(os.environ, AMD_LOG_LEVEL="4")
    p = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True, env=env, timeout=600)
    err = p.stderr
    seg = err[err.find("MARK-BEGIN"):err.find("MARK-END")] if "MARK-BEGIN" in err else err
    heads = collections.Counter()
    for line in seg.splitlines():
        m = re.search(r"Dispatch Header = (0x[0-9a-fA-F]+) \(type=(\d+), barrier=(\d+), acquire=(\d+), release=(\d+)\)", line)
        if m:
            heads[(m.group(1), f"barrier={m.group(3)} acquire={m.group(4)} release={m.group(5)}")] += 1
    other = collections.Counter()
    for line in seg.splitlines():
        for key in ("Barrier Header", "barrier packet", "BarrierAnd", "Marker", "hsa_signal"):
            if key.lower() in line.lower():
                other[key] += 1
    out.append(f"## {mode}: rc {p.returncode}, {len(seg.splitlines())} log lines between the marks")
    for (h, what), n in heads.most_common():
        out.append(f"- {n} x dispatch header {h}: {what}  (fence scope 0 = none, 1 = agent, 2 = system)")
    if not heads:
        sample = [l for l in seg.splitlines() if "ispatch" in l][:5]
        out.append("- no 'Dispatch Header' lines; sample: " + " | ".join(s[:160] for s in sample))
    for k, n in other.items():
        out.append(f"- {n} log lines mentioning {k}")
text = "\n".join(out)
print(text)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/fence_scope.md", "w").write(text + "\n")
