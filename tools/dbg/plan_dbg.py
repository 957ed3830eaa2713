import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
from hydrainfer_amd import launch_plan
DEV = torch.device("cuda:0")
sh = LlamaShape(4096, 11008, 3, 32, 32, 128, 32064)
res = {}
for ex in sys.argv[1:] or ["graph", "plan-nochain", "plan"]:
    model = LlamaForCausalLM.random_init(sh, torch.float16, DEV, seed=3)
    r = DecodeRunner(model, RunnerConfig(batch=32, prompt_len=40, n_generate=40, use_graph=True, executor=ex), seed=4)
    g = torch.Generator().manual_seed(0)
    r.prefill(torch.randint(5, sh.vocab_size - 1, (32, 40), generator=g).to(DEV))
    for i in range(8):
        r.step()
        torch.cuda.synchronize()
        s = model.xreg_sync
        print(ex, "step", i, "sync word1 sum", int(s[:, :, 1].abs().sum()), "word0", s[:, :, 0].flatten().tolist(),
              "plan err", int(r.graph.error_word[0]) if isinstance(r.graph, launch_plan.LaunchPlan) else None, flush=True)
    if isinstance(r.graph, launch_plan.LaunchPlan):
        print(ex, "launches", r.graph.n_launches, "any-order", r.graph.n_any_order)
    res[ex] = (torch.stack(r.tokens).cpu(), r.pool.clone())
for ex in res:
    print(ex, "tokens equal graph:", torch.equal(res[ex][0], res["graph"][0]) if "graph" in res else None,
          "pool equal:", torch.equal(res[ex][1], res["graph"][1]) if "graph" in res else None)
