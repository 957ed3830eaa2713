"""Debug harness for the multi-process E/P/D engine: run under torch.distributed.run with N ranks
(HX_SINGLE_DEVICE=1 HX_DIST_BACKEND=gloo puts them all on cuda:0).  Prints per-rank progress and
dumps every thread's stack if a rank makes no progress for 60 s."""
import argparse
import faulthandler
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="tiny")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--max-tokens", type=int, default=16)
    args = ap.parse_args()
    faulthandler.dump_traceback_later(60, exit=True)
    import bench
    from hydrainfer_amd import parallel
    from hydrainfer_amd.model.llama import LlamaForCausalLM
    ctx = parallel.init_from_env()
    rank = ctx.rank
    dev = torch.device("cuda:0" if os.environ.get("HX_SINGLE_DEVICE") == "1" else f"cuda:{ctx.local_rank}")
    torch.cuda.set_device(dev)
    log = lambda *a: print(f"[rank {rank} {time.perf_counter():.3f}]", *a, file=sys.stderr, flush=True)
    shape, name = bench.model_shape(args.model)
    dtype = torch.bfloat16
    model = LlamaForCausalLM.random_init(shape, dtype, dev, seed=0)
    vision, pixels = bench.make_vision(shape, dtype, dev)
    engine = bench.build_rank_engine(ctx, model, vision, shape, dtype, dev, args.batch, 128, args.max_tokens)
    log("engine built", engine.roles[rank])
    from hydrainfer_amd._C.data_transfer import block_migration as bm
    node = engine.node
    mine = {"kv": node.kv_cache_block_manager.memory_handle if node.kv_cache_block_manager else None,
            "image": node.image_cache_block_manager.memory_handle if node.image_cache_block_manager else None}
    pools = ctx.all_gather_object(mine)
    for r, role in enumerate(engine.roles):
        if r == rank:
            continue
        if node.node_type.enable_prefill and "E" in role and pools[r]["image"]:
            bm._open(pools[r]["image"]); log("opened image pool of", r)
        if node.node_type.enable_decode and "P" in role and pools[r]["kv"]:
            bm._open(pools[r]["kv"]); log("opened kv pool of", r)
    real_step = node.step
    state = {"n": 0}

    def step():
        n = real_step()
        state["n"] += 1
        if n or state["n"] % 2000 == 0:
            s = node.batch_scheduler
            log(f"step {state['n']}: batch {n} running {len(s.running)} waiting {len(s.waiting)} "
                f"migrating {s.migrating_cnt} finished {len(node.finished)} held {len(engine.held)}")
        return n
    node.step = step
    res = bench.measure_disaggregated(ctx, engine, shape, dev, pixels, args.batch, 128, args.max_tokens)
    if rank == 0:
        print(res)
    faulthandler.cancel_dump_traceback_later()
    ctx.shutdown()


if __name__ == "__main__":
    main()
