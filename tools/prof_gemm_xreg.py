#!/usr/bin/env python3
"""Target program for rocprofv3 PMC passes (FETCH_SIZE | WRITE_SIZE, separate passes) over the
activations-in-registers decode GEMM as the 7B decode step launches it at M = 32: gate|up with the
fused silu*mul epilogue, down, qkv — three cold launches each (a 300 MB fill between launches evicts
the 256 MiB Infinity Cache).
    python3 tools/prof_gemm_xreg.py                      (under rocprofv3 --pmc ...)
    python3 tools/prof_gemm_xreg.py summarize <fetch dir> <write dir> <out.json>"""
import csv, glob, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {"gate_up+silu": (22016, 4096, "<BF16,2,32,1,0,0>"), "down": (4096, 11008, "<BF16,2,22,0,0,0>"),
          "qkv": (12288, 4096, "<BF16,2,32,0,0,0>"),
          "norm+gate_up+silu": (22016, 4096, "<BF16,2,32,1,0,1>"), "norm+qkv": (12288, 4096, "<BF16,2,32,0,0,1>")}
M = 32


def run():
    import torch
    from hydrainfer_amd._C.kernel import gemm
    dev, dt = torch.device("cuda:0"), torch.bfloat16
    big = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)
    from hydrainfer_amd._C.kernel import gemm as G
    for name, (N, K, _) in SHAPES.items():
        fused = "gate_up" in name
        nrm = name.startswith("norm+")
        pk = [gemm.pack_weight_xreg((torch.randn((N, K), device=dev) * 0.02).to(dt), interleave_halves=fused) for _ in range(3)]
        xf = gemm.to_fragment_major(torch.randn((M, K), device=dev).to(dt))
        ws = torch.empty(max(gemm.xreg_workspace_floats(M, N, K), 1), dtype=torch.float32, device=dev)
        act = torch.empty(gemm.fragment_major_elems(M, N // 2), dtype=dt, device=dev)
        slabs = torch.randn((4, M, K), device=dev)
        resid = torch.randn((M, K), device=dev).to(dt)
        nw = torch.randn(K, device=dev).to(dt)
        sync = torch.zeros((3, G.XREG_SYNC_WORDS), dtype=torch.int32, device=dev)
        for i in range(3):
            big.fill_(i)
            if nrm and fused:
                gemm.norm_gate_up_silu_xreg(resid, slabs, 4, nw, 1e-5, xf, pk[i], N // 2, act, sync[i])
            elif nrm:
                gemm.norm_linear_decode_xreg(resid, slabs, 4, nw, 1e-5, xf, pk[i], N, ws, sync[i])
            elif fused:
                gemm.gate_up_silu_xreg(xf, pk[i], N // 2, act, frag_shape=(M, K))
            else:
                gemm.linear_decode_partial_xreg(xf, pk[i], N, ws, frag_shape=(M, K))
        torch.cuda.synchronize()
        del pk


def counters(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_xreg_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                key = r["Kernel_Name"].split("gemm_xreg_kernel")[1].split("(")[0].replace("hx::", "").replace(" ", "")
                parts = key.strip("<>").split(",")      # <T, MB, KW, EPI, DBG, NORM[, RM]>: defaults appended by round
                parts = (parts + ["0"] * 6)[:6] if len(parts) <= 6 else parts[:6]
                key = "<" + ",".join(parts) + ">"
                out.setdefault(key, []).append(float(r["Counter_Value"]))
    return out


def summarize(dir_f, dir_w, out):
    fe, wr = counters(dir_f, "FETCH_SIZE"), counters(dir_w, "WRITE_SIZE")
    res = {"kernel": "gemm_xreg_kernel (activations in registers, M=32, bf16, LLaVA-1.5-7B shapes, fragment-major x)",
           "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 tools/prof_gemm_xreg.py "
                      "(separate passes; a 300 MB fill between launches evicts the Infinity Cache)",
           "correction": "gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads "
                         "(MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact; both in KiB",
           "shapes": []}
    for name, (N, K, cfg) in SHAPES.items():
        f = statistics.median(fe[cfg]) * 1024 * 2
        w = statistics.median(wr[cfg]) * 1024
        wbytes = N * K * 2
        xbytes = M * K * 2
        if "gate_up" in name:
            out_bytes, what = M * (N // 2) * 2, "act (bf16, fragment-major)"
        else:
            splits = 4 if K == 11008 else 1
            out_bytes, what = splits * M * N * 4, f"{splits} fp32 slab(s)"
        if name.startswith("norm+"):   # + the residual rows rewritten in place and x (fragment-major, written through)
            out_bytes += 2 * M * K * 2
            what += " + residual + x"
        if cfg not in fe or cfg not in wr:
            continue
        res["shapes"].append({"name": name, "config": cfg, "N": N, "K": K,
                              "fetch_bytes_corrected": f, "algorithmic_weight_bytes": wbytes,
                              "fetch_over_weights": round(f / wbytes, 4),
                              "x_bytes_once": xbytes, "write_bytes": w, "output_bytes_expected": out_bytes,
                              "output": what, "write_over_output": round(w / out_bytes, 4)})
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summarize":
        summarize(*sys.argv[2:5])
    else:
        run()
