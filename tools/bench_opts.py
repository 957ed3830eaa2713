#!/usr/bin/env python3
"""bench.py's decode step under hx_debug_set_option knobs, same process / same box (A/B of tuning options):
    python tools/bench_opts.py "decode_waves=8" "decode_nt=0" ...      each argument is one configuration
(comma-separated name=value pairs; "" = defaults).  Prints ms per step (median of 3 x 40 graph replays)."""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hydrainfer_amd import _lib
from hydrainfer_amd.model.llama import LlamaForCausalLM
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig

dev, dt = torch.device("cuda:0"), torch.bfloat16
shape, _ = bench.model_shape(os.environ.get("MODEL", "7b"))
model = LlamaForCausalLM.random_init(shape, dt, dev, seed=0)
lib = _lib.lib()
for cfg in (sys.argv[1:] or [""]):
    pairs = [p.split("=") for p in cfg.split(",") if p]
    for k, v in pairs:
        assert lib.hx_debug_set_option(k.encode(), int(v)) == 0, k
    r = DecodeRunner(model, RunnerConfig(batch=32, prompt_len=704, n_generate=256, use_graph=True), seed=1)
    r.fake_prefill() if hasattr(r, "fake_prefill") else r.prefill(torch.randint(5, 30000, (32, 704), device=dev))
    for _ in range(5):
        r.step(record=False)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            r.step(record=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print(f"{cfg or 'defaults':40s} {statistics.median(ts):.4f} ms/step", flush=True)
    for k, v in pairs:   # back to defaults: the known ones
        lib.hx_debug_set_option(k.encode(), {"decode_waves": 4, "decode_nt": 1, "xreg_stagger": 1}.get(k, 0))
    del r
