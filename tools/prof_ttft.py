#!/usr/bin/env python3
"""One image+text request on an idle replica, eager (no graphs), N times — target program for a
rocprofv3 kernel trace of the TTFT path (CLIP encode + projector + 704-token prefill + sample)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hydrainfer_amd.model.llama import LlamaForCausalLM
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig

dev = torch.device("cuda:0")
dtype = torch.bfloat16
shape, _ = bench.model_shape("7b")
model = LlamaForCausalLM.random_init(shape, dtype, dev, seed=0)
runner = DecodeRunner(model, RunnerConfig(batch=1, prompt_len=704, n_generate=8, use_graph=False), seed=0)
vision, pixels = bench.make_vision(shape, dtype, dev)
pixels = pixels.to(dev)
prompts = bench.synth_prompts(1, 704, shape.vocab_size, dev)
for i in range(6):
    feats = vision(pixels)
    first = runner.prefill(prompts, feats, 32000, requests=[0])
    first[0].item()
torch.cuda.synchronize()
print("done")
