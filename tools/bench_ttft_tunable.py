#!/usr/bin/env python3
"""Eager single-request prefill time (704 tokens, 7B) — run once plain and once with
PYTORCH_TUNABLEOP_ENABLED=1 to see what tuned library GEMMs would buy the TTFT path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hydrainfer_amd.model.llama import LlamaForCausalLM
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig

if os.environ.get("HX_BLAS"):          # "cublas" (= rocBLAS) | "cublaslt" (= hipBLASLt)
    torch.backends.cuda.preferred_blas_library(os.environ["HX_BLAS"])
dev = torch.device("cuda:0")
dtype = torch.bfloat16
shape, _ = bench.model_shape(sys.argv[1] if len(sys.argv) > 1 else "7b")
model = LlamaForCausalLM.random_init(shape, dtype, dev, seed=0)
runner = DecodeRunner(model, RunnerConfig(batch=1, prompt_len=704, n_generate=8, use_graph=True), seed=0)
prompts = bench.synth_prompts(1, 704, shape.vocab_size, dev)
feats = torch.zeros(1, 576, shape.hidden_size, dtype=dtype, device=dev)
if os.environ.get("HX_TUNE"):          # the engine's own start-up pass (serve.tune_library_gemms), tuning off again behind it
    from hydrainfer_amd.engine.serve import tune_library_gemms
    rot = os.environ.get("HX_TUNE_ROT")
    print("tune_library_gemms:", tune_library_gemms(model, rows=(704,), rotating_buffer_mb=int(rot) if rot else None))
t0 = time.perf_counter()
for _ in range(3):
    runner.prefill(prompts, feats, 32000, requests=[0])[0].item()
print("warm-up (incl. any tuning) s:", round(time.perf_counter() - t0, 1))
pg, p_ids, p_feats, p_first = runner.capture_prefill(0, 32000)
p_ids.copy_(prompts[0])
ts = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pg.replay(); p_first[0].item()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("graph-replayed 704-token prefill p50 ms:", round(ts[len(ts) // 2], 3), "tunable:", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"),
      "blas:", torch.backends.cuda.preferred_blas_library())
