#!/usr/bin/env python3
"""Long fuzz of the persistent prefill kernel against the per-item one (bit-identical by construction): random ragged
batches, head counts, head sizes, block sizes, cached prefixes, causal or not, every deal mode."""
import math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd
DEV = "cuda:0"
lib = _lib.lib()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    dt = rnd.choice((torch.float16, torch.bfloat16))
    D = rnd.choice((64, 128))
    H, HK = rnd.choice(((8, 8), (8, 2), (5, 1), (16, 4), (32, 32), (3, 3), (24, 8)))
    bs = rnd.choice((16, 32, 64))
    B = rnd.choice((1, 2, 3, 5, 9, 17, 40, 130))
    longest = rnd.choice((130, 300, 704, 1500)) if B < 40 else rnd.choice((130, 260))
    q_lens = [rnd.choice((0, 1, rnd.randint(2, 64), rnd.randint(65, longest), 128, 129, 256)) for _ in range(B)]
    q_lens[rnd.randrange(B)] = longest
    kv_lens = [ql + rnd.choice((0, 0, rnd.randint(1, 200), 16 * rnd.randint(1, 9))) if ql else rnd.randint(0, 30) for ql in q_lens]
    causal = rnd.random() < 0.8
    g = torch.Generator().manual_seed(case)
    nblk = sum((l + bs - 1) // bs for l in kv_lens) + 5
    kc = torch.randn((nblk, bs, HK, D), generator=g).to(dt).to(DEV)
    vc = torch.randn((nblk, bs, HK, D), generator=g).to(dt).to(DEV)
    perm = torch.randperm(nblk, generator=g).tolist()
    tables, cu_b, cu_q, cu_k, used = [], [0], [0], [0], 0
    for ql, kl in zip(q_lens, kv_lens):
        nb = (kl + bs - 1) // bs
        tables += perm[used: used + nb]; used += nb
        cu_b.append(cu_b[-1] + nb); cu_q.append(cu_q[-1] + ql); cu_k.append(cu_k[-1] + kl)
    q = torch.randn((cu_q[-1], H, D), generator=g).to(dt).to(DEV)
    i32 = lambda x: torch.tensor(x if x else [0], dtype=torch.int32, device=DEV)
    args = (kc, vc, i32(cu_q), i32(cu_k), i32(tables), i32(cu_b), None, max(q_lens), max(max(kv_lens), 1), 1 / math.sqrt(D), 0.0, -1, 0 if causal else -1, 0)
    outs = {}
    for name, opts in (("item", {"fwd_persistent": 0}), ("tiles", {"fwd_persistent": 2, "fwd_units": 0}),
                       ("units", {"fwd_persistent": 2, "fwd_units": 1, "fwd_seq_group": 1}),
                       ("g4", {"fwd_persistent": 2, "fwd_seq_group": 4}), ("auto", {})):
        for k, v in {"fwd_persistent": 1, "fwd_units": -1, "fwd_seq_group": 0, **opts}.items():
            lib.hx_debug_set_option(k.encode(), v)
        out = torch.full_like(q, float("nan"))
        mha_varlen_fwd(out, q, *args)
        torch.cuda.synchronize()
        outs[name] = out
    ok = all(torch.equal(outs["item"], o) for o in outs.values()) and bool(torch.isfinite(outs["item"].float()).all())
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {dt} D={D} H={H}/{HK} bs={bs} B={B} causal={causal} q={q_lens[:8]}.. kv={kv_lens[:8]}..",
              {k: float((outs['item'].float() - o.float()).abs().max()) for k, o in outs.items()}, flush=True)
for k, v in {"fwd_persistent": 1, "fwd_units": -1, "fwd_seq_group": 0}.items():
    lib.hx_debug_set_option(k.encode(), v)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
