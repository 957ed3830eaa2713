#!/usr/bin/env python3
"""Who owns the single-request TTFT (round-4 review, item 5): the TTFT leg of bench.py — CLIP encode + projector, then the
704-token prefill + greedy sample, each replayed from a hipGraph on an idle replica — with a marker launch (hx_memset_zero
of 4 bytes) in front of, between and behind the two phases, so that a rocprofv3 kernel trace of this program can be cut
into encode / prefill and, inside each, library GEMM / attention / elementwise / idle gaps.

    run:      rocprofv3 --kernel-trace -d <dir> -o ttft -- python3 tools/ttft_timeline.py run
    analyse:  python3 tools/ttft_timeline.py analyse <dir>/ttft_results.db [out.md]
"""
import os, sqlite3, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import time
    import torch
    import bench
    from hydrainfer_amd import _lib
    from hydrainfer_amd.model.llama import LlamaForCausalLM
    from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    shape, name = bench.model_shape("7b")
    dt = torch.bfloat16
    model = LlamaForCausalLM.random_init(shape, dt, dev, seed=0)
    model.prepare_decode(max_rows=32, keep_row_major=True)
    runner = DecodeRunner(model, RunnerConfig(batch=32, prompt_len=704, n_generate=256), seed=0)
    vision, pixels = bench.make_vision(shape, dt, dev)
    prompts = bench.synth_prompts(32, 704, shape.vocab_size, dev)
    itid = bench.image_token_id(shape.vocab_size)
    pixels = pixels.to(dev)
    # warm every kernel / GEMM heuristic, then capture both phases
    runner.prefill(prompts, vision(pixels).expand(32, -1, -1), itid, requests=[0])
    pg, p_ids, p_feats, p_first = runner.capture_prefill(0, itid)
    p_ids.copy_(prompts[0])
    static_pixels = pixels.clone()
    s = torch.cuda.Stream(device=dev); s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        vision(static_pixels)
    torch.cuda.current_stream(dev).wait_stream(s)
    vg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(vg):
        v_out = vision(static_pixels)
    marker = torch.zeros(1, dtype=torch.int32, device=dev)
    ts = []
    for i in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.memset_zero(marker)
        static_pixels.copy_(pixels)
        vg.replay()
        _lib.memset_zero(marker)
        p_feats.copy_(v_out[0])
        pg.replay()
        _lib.memset_zero(marker)
        p_first[0].item()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("ttft ms per iteration (host clock, markers included):", [round(t, 3) for t in ts])


def kind(n):
    if "Cijk" in n:
        return "library GEMM (hipBLASLt Cijk_*)"
    if "attn_fwd" in n:
        return "attention (attn_fwd*)"
    if "zero_kernel" in n:
        return "marker"
    return "elementwise / norm / rope / copies"


def analyse(db, out_path=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "zero_kernel" in r[0]]
    # iterations = consecutive marker triples whose span is below 100 ms
    its = []
    k = 0
    while k + 2 < len(marks):
        a, b, d = marks[k], marks[k + 1], marks[k + 2]
        if rows[d][1] - rows[a][1] < 100e6 and b - a > 50 and d - b > 50:
            its.append((a, b, d)); k += 3
        else:
            k += 1
    its = its[2:] if len(its) > 4 else its            # drop the first two (clocks, caches)
    out = [f"# Single-request TTFT by owner — {len(its)} iterations of tools/ttft_timeline.py (7B, bf16, 1 image + 704-token prompt, "
           "both phases replayed from hipGraphs), rocprofv3 kernel trace, medians", ""]
    total = statistics.median((rows[d][1] - rows[a][2]) / 1e3 for a, b, d in its)
    out.append(f"GPU span marker to marker: **{total / 1e3:.3f} ms** (the host adds the pixel copy, two graph launches and the token's D2H copy)\n")
    out.append("| phase | owner | launches | busy us | share of the span |\n|---|---|---|---|---|")
    for label, lo, hi in (("encode (23 CLIP layers + projector)", 0, 1), ("prefill (704 tokens, 32 layers) + sample", 1, 2)):
        per = {}
        gaps, spans = [], []
        for it in its:
            a, b = it[lo], it[hi]
            seg = rows[a + 1:b]
            spans.append((rows[b][1] - rows[a][2]) / 1e3)
            g_ = 0.0
            prev = rows[a][2]
            acc = {}
            for n, s_, e_ in seg:
                kd = kind(n)
                acc.setdefault(kd, [0, 0.0])
                acc[kd][0] += 1; acc[kd][1] += (e_ - s_) / 1e3
                g_ += max(0, s_ - prev) / 1e3
                prev = e_
            g_ += max(0, rows[b][1] - prev) / 1e3
            gaps.append(g_)
            for kd, v in acc.items():
                per.setdefault(kd, []).append(v)
        for kd, vs in sorted(per.items(), key=lambda kv: -statistics.median(v[1] for v in kv[1])):
            busy = statistics.median(v[1] for v in vs)
            out.append(f"| {label} | {kd} | {vs[0][0]} | {busy:.1f} | {100 * busy / total:.1f} % |")
        out.append(f"| {label} | idle gaps between kernels | | {statistics.median(gaps):.1f} | {100 * statistics.median(gaps) / total:.1f} % |")
        out.append(f"| {label} | **phase total** | | **{statistics.median(spans):.1f}** | {100 * statistics.median(spans) / total:.1f} % |")
    # top kernels of the prefill phase
    a, b, d = its[len(its) // 2]
    top = {}
    for n, s_, e_ in rows[b + 1:d]:
        key = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:90]
        top.setdefault(key, [0, 0.0]); top[key][0] += 1; top[key][1] += (e_ - s_) / 1e3
    out.append("\nTop kernels of the prefill phase (one iteration):\n\n| kernel | launches | total us |\n|---|---|---|")
    for key, v in sorted(top.items(), key=lambda kv: -kv[1][1])[:10]:
        out.append(f"| `{key}` | {v[0]} | {v[1]:.1f} |")
    text = "\n".join(out)
    print(text)
    if out_path:
        open(out_path, "w").write(text + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "analyse":
        analyse(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        run()
