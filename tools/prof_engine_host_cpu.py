#!/usr/bin/env python3
"""Host time of ONE engine decode step, with the device taken out: the node's real scheduler, step loop, executor and
decode stager (engine/graph_decode.py::DecodeStager) in front of a stand-in for the device side (launch = stage into a
numpy buffer, fetch = constant tokens).  No GPU needed — this is the number the engine's single host thread has to stay
under the GPU's step time with (round-5 review, item 4: < 150 us at 64 rows).
    python tools/prof_engine_host_cpu.py [rows=64] [prompt=704] [--profile]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS

import numpy as np
import torch

from hydrainfer_amd.engine import BatchSchedulerConfig, InstructionCreator, SamplingParameters, TokenRequest
from hydrainfer_amd.engine.graph_decode import DecodeStager
from hydrainfer_amd.engine.node import LocalCluster
from hydrainfer_amd.engine.serve import quiet_gc
from tests.engine_util import CpuPoolManager, make_node

rows_n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
prompt = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 704
BS, CAP = 16, (prompt + 256 + 15) // 16 + 1


class HostOnlyDecoder:
    """GraphedDecoder's interface with the device side removed."""

    def __init__(self, kv, max_batch):
        self.pad_to, self.max_batch, self.cap = 4, max_batch, CAP
        self.stager = DecodeStager(max_batch, CAP, BS, 0, 4096, 32064)
        self.st = [np.zeros(self.stager.staging_words, dtype=np.int32) for _ in range(2)]
        self.launches, self.rows_of = 0, {}

    def fits(self, n, max_blocks):
        return n <= self.max_batch and max_blocks <= self.cap

    def launch(self, rows):
        self.launches += 1
        B = (len(rows) + 3) // 4 * 4
        self.stager.stage(self.st[self.launches % 2], rows, B)
        self.rows_of[self.launches] = len(rows)
        return self.launches

    def launch_cohort(self, n, pos, slots, starts, grown):
        self.launches += 1
        B = (n + 3) // 4 * 4
        self.stager.stage_cohort(self.st[self.launches % 2], n, B, pos, slots, starts, grown)
        self.rows_of[self.launches] = n
        return self.launches

    def fetch(self, launch_id):
        return [7] * self.rows_of.pop(launch_id)

    def warmup(self, *a, **k):
        pass


shape = NS(num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1, head_dim=8, max_position_embeddings=4096)


class LM:
    image_token_id = 32000
    language_model = NS(shape=shape)

    def forward(self, ids, feats, pos, params):
        return torch.full((params.selected_token_ids.numel(),), 7, dtype=torch.int64)


kv = CpuPoolManager(1, 2, rows_n * CAP + 8, BS, 1, 8)
cfg = BatchSchedulerConfig(priority="prefill", max_running_requests=rows_n, chunked_prefill=True, token_budgets=2048, image_budgets=8)
node = make_node("EPD0", "EPD", LM(), None, kv, None, shape, torch.float32, torch.device("cpu"), cfg)
node.executor.fill_executor.graph_decoder = HostOnlyDecoder(kv, rows_n)
cluster = LocalCluster([node])
creator = InstructionCreator(32000, 576, BS)
g = torch.Generator().manual_seed(0)
for i in range(rows_n):
    cluster.add_request(creator.process(TokenRequest(i, torch.randint(1000, 31999, (prompt,), generator=g).tolist(), None, (336, 336), i,
                                                     SamplingParameters(max_tokens=256))))
with quiet_gc():
    n_prefill = (rows_n * prompt + 2047) // 2048 + 2          # (the cohort keeps a request's token list for its end: count steps)
    for _ in range(n_prefill + 10):
        cluster.step()
    assert len(node.batch_scheduler.running) == rows_n and not node.batch_scheduler.waiting
    ts = []
    prof = cProfile.Profile() if "--profile" in sys.argv else None
    for i in range(200):
        if prof and i == 100:
            prof.enable()
        t0 = time.perf_counter()
        cluster.step()
        ts.append(time.perf_counter() - t0)
    if prof:
        prof.disable()
ts.sort()
fe = node.executor.fill_executor
print(f"cohort steps {fe.n_cohort_steps}; " f"{rows_n} rows, prompt {prompt}: host time of a decode step: median {ts[len(ts) // 2] * 1e6:.0f} us, p90 {ts[int(len(ts) * 0.9)] * 1e6:.0f} us, "
      f"min {ts[0] * 1e6:.0f} us")
if prof:
    pstats.Stats(prof).sort_stats("tottime").print_stats(22)
