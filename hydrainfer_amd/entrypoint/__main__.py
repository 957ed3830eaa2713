"""python -m hydrainfer_amd.entrypoint [--model 7b|13b|tiny] [--host H] [--port P] [--max-running N] [--tokenizer DIR]

One collocated EPD replica on cuda:0 behind the OpenAI-compatible endpoint (hydrainfer/entrypoint/entrypoint.py serves
the same routes through Ray + zmq).  No checkpoints offline: LLaVA-1.5-shaped RANDOM weights (what bench.py measures) and
the synthetic tokenizer unless --tokenizer names a checkpoint directory.  The reference's benchmark client
(benchmark/benchmark.py --backend ours --base-url http://H:P/v1) drives it unchanged."""
import argparse

import torch


def main():
    ap = argparse.ArgumentParser(prog="python -m hydrainfer_amd.entrypoint")
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8888)
    ap.add_argument("--max-running", type=int, default=64)
    ap.add_argument("--max-tokens", type=int, default=1024, help="upper bound of prompt + generated tokens per request (cache sizing)")
    ap.add_argument("--tokenizer", default=None, help="checkpoint directory for transformers.AutoTokenizer (default: synthetic)")
    ap.add_argument("--tune-rows", default="", help="comma-separated prompt lengths to autotune the library's prefill GEMMs for at start-up "
                    "(serve.tune_library_gemms; opt-in: the pass runs every candidate kernel of the library — read its caution)")
    a = ap.parse_args()
    from hydrainfer_amd import _lib
    from hydrainfer_amd.engine.node import LocalCluster
    from hydrainfer_amd.engine.request_processor import InstructionCreator
    from hydrainfer_amd.engine.scheduler import BatchSchedulerConfig
    from hydrainfer_amd.engine.serve import build_node, tune_library_gemms, warm_library_gemms
    from hydrainfer_amd.entrypoint import ApiServer, EngineFrontend, HFTokenizer, SyntheticTokenizer
    from hydrainfer_amd.model.clip import CLIP_VIT_L_14_336, ClipShape, LlavaVisionModel
    from hydrainfer_amd.model.llama import LLAVA_1_5_7B, LLAVA_1_5_13B, LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.llava import LlavaLanguageModel
    from hydrainfer_amd.model.processor import ClipImageProcessor
    import dataclasses
    _lib.lib()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    shape = {"7b": LLAVA_1_5_7B, "13b": LLAVA_1_5_13B, "tiny": LlamaShape(256, 512, 2, 2, 2, 128, 32064)}[a.model]
    cshape = (dataclasses.replace(CLIP_VIT_L_14_336, projector_hidden_size=shape.hidden_size) if a.model != "tiny" else
              ClipShape(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=336,
                        patch_size=14, projector_hidden_size=shape.hidden_size))
    itid = 32000
    lm = LlavaLanguageModel(LlamaForCausalLM.random_init(shape, dt, dev, seed=0), image_token_id=itid)
    vision = LlavaVisionModel.random_init(cshape, dt, dev, seed=1)
    sched = BatchSchedulerConfig(priority="prefill", max_running_requests=a.max_running, chunked_prefill=True,
                                 token_budgets=4096, image_budgets=8)
    per_req = (a.max_tokens + 15) // 16 + 1
    node = build_node("EPD0", "EPD", lm, vision, shape, dt, dev, kv_blocks=a.max_running * per_req + 64,
                      image_blocks=2 * a.max_running, n_image_tokens=576, sched=sched, max_blocks_per_seq=per_req)
    warm_library_gemms(lm, sched.token_budgets, a.max_running)
    if a.tune_rows:
        print("library GEMMs tuned:", tune_library_gemms(lm, rows=tuple(int(r) for r in a.tune_rows.split(","))), flush=True)
    creator = InstructionCreator(image_token_id=itid, n_image_tokens_per_image=576, block_size=16,
                                 max_position_embeddings=shape.max_position_embeddings)
    tok = HFTokenizer(a.tokenizer) if a.tokenizer else SyntheticTokenizer(image_token_id=itid)
    front = EngineFrontend(LocalCluster([node]), creator, device=dev)
    print(f"serving {a.model} ({a.dtype}, random weights) on http://{a.host}:{a.port}/v1/chat/completions", flush=True)
    ApiServer(front, tok, ClipImageProcessor(), host=a.host, port=a.port).run()


if __name__ == "__main__":
    main()
