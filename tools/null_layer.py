#!/usr/bin/env python3
"""The ceiling of the decode step's LAUNCH STRUCTURE (round-4 review, "next round" item 1a): the step replayed with every
launch of a layer replaced by a math-free read of the same bytes on the same grid (bench.null_step_object), next to the
built step on the same model, KV pool and block table in the same process, interleaved.  Writes a markdown table.

    python tools/null_layer.py [--model 7b|13b] [--steps 20] [--rounds 5] [--out gpurun_out/null_layer.md]
Under rocprofv3 --kernel-trace the per-launch times of both forms are in the trace (tools/layer_timeline.py)."""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hydrainfer_amd.model.llama import LlamaForCausalLM
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
from hydrainfer_amd import parallel

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="7b")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--out", default="gpurun_out/null_layer.md")
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
shape, name = bench.model_shape(a.model)
dt = torch.bfloat16
model = LlamaForCausalLM.random_init(shape, dt, dev, seed=0)
model.prepare_decode(max_rows=32, keep_row_major=False)
cfg = RunnerConfig(batch=32, prompt_len=704, n_generate=256, use_graph=True, executor="plan")
runner = DecodeRunner(model, cfg, seed=0)
ctxs = bench.timed_contexts(704, 256, a.steps)
ctx = parallel.init_from_env()
runner.input_ids.copy_(torch.randint(1000, 30000, (32,), device=dev))
built, null, best = [], [], []
for r in range(a.rounds):
    el = bench.decode_leg(ctx, model, runner, ctxs, 2, 704)
    ms = el / len(ctxs) * 1e3
    o = bench.null_step_object(model, runner, ctxs, ms, reps=3)
    built.append(ms); null.append(o["ms_per_step"]); best.append(o["null_step_best_grid_ms"])
    print(f"round {r}: built {ms:.4f} ms  null {o['ms_per_step']:.4f}  null(512 wgs) {o['null_step_best_grid_ms']:.4f}", flush=True)
step_bytes = sum(runner.step_bytes(c * 32) for c in ctxs) / len(ctxs)
med = statistics.median
fr = lambda ms: step_bytes / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS
rs = bench.read_stream_ceiling_gbs(dev)
lines = [f"# Null layer: the launch structure's ceiling — {name}, batch 32, {bench.ctx_label(ctxs)}", "",
         f"Same process, same weights / KV pool / block table, {a.rounds} interleaved rounds (median); algorithmic bytes per step "
         f"{step_bytes / 1e9:.3f} GB; read-stream probe of this box {rs:.0f} GB/s.", "",
         "| form | ms per step | of 8 TB/s | built / this |", "|---|---|---|---|",
         f"| built step (5 launches per layer, arithmetic, hand-overs) | {med(built):.4f} | {fr(med(built)):.4f} | 1 |",
         f"| null step, real grids (256 workgroups per weight launch, 1024 for the paged read) | {med(null):.4f} | {fr(med(null)):.4f} | {med(null) / med(built):.4f} |",
         f"| null step, the read probe's best grid (512 workgroups per weight launch) | {med(best):.4f} | {fr(med(best)):.4f} | {med(best) / med(built):.4f} |",
         "", "rounds (built, null, null best grid): " + "; ".join(f"{b:.4f} / {n:.4f} / {q:.4f}" for b, n, q in zip(built, null, best))]
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
open(a.out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
