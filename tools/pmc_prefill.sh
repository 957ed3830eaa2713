# PMC passes + kernel trace over the prefill attention kernel (4 x 704 tokens): `bash tools/pmc_prefill.sh` on the GPU box.
# Writes gpurun_out/r4/attn_prefill_pmc_b$HX_PREFILL_B.json (medians over the 12 launches of tools/prof_attn_prefill32.py;
# HX_PREFILL_B = number of 704-token sequences, default 4).
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repo root (gpurun exports it)}
test -f "$R/tools/prof_attn_prefill32.py" || { echo "no tools/prof_attn_prefill32.py under $R" >&2; exit 2; }
cd /tmp; export TMPDIR=/tmp
export HX_PREFILL_B=${HX_PREFILL_B:-4}
O=$R/gpurun_out/r4
mkdir -p "$O"
rm -rf "$O"/pmc_prefill_*
i=0
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $O/pmc_prefill_$i -o p -- python3 $R/tools/prof_attn_prefill32.py > $O/pmc_prefill_$i.log 2>&1 \
    || { echo "PMC pass $i failed:" >&2; tail -5 $O/pmc_prefill_$i.log >&2; exit 1; }
done
rocprofv3 --kernel-trace --output-format csv -d $O/pmc_prefill_trace -o p -- python3 $R/tools/prof_attn_prefill32.py > $O/pmc_prefill_trace.log 2>&1 \
  || { echo "kernel trace failed:" >&2; tail -5 $O/pmc_prefill_trace.log >&2; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, collections, statistics, json, os
B = int(os.environ.get("HX_PREFILL_B", "4"))
med = {}
under = []
for i in (1, 2, 3):
    fs = glob.glob(f"gpurun_out/r4/pmc_prefill_{i}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "attn_fwd32" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if i == 1 and r["Counter_Name"] == "SQ_WAVES" and "Start_Timestamp" in r:
                under.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in acc.items():
        med[k] = statistics.median(v)
fs = glob.glob("gpurun_out/r4/pmc_prefill_trace/**/*kernel_trace.csv", recursive=True)
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(fs[0])) if "attn_fwd32" in r["Kernel_Name"]]
flops = 4 * 32 * 128 * B * (704 * 705 // 2)
cyc = med["GRBM_GUI_ACTIVE"] / 8
out = {"sequences_of_704_tokens": B, "launches": len(dur), "duration_us_kernel_trace_median": statistics.median(dur), "counters_median": med,
       "kernel_cycles": cyc, "mfma_busy_frac": med["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
       "mfma_cycles_with_valu_coexecuting_frac": med["SQ_VALU_MFMA_COEXEC_CYCLES"] / med["SQ_VALU_MFMA_BUSY_CYCLES"],
       "lds_bank_conflict_frac_of_lds_cycles": med["SQ_LDS_BANK_CONFLICT"] / med["SQ_LDS_IDX_ACTIVE"],
       "valu_per_mfma": med["SQ_INSTS_VALU"] / med["SQ_INSTS_MFMA"], "salu_per_mfma": med["SQ_INSTS_SALU"] / med["SQ_INSTS_MFMA"],
       "wait_any_frac_of_wave_cycles": med["SQ_WAIT_ANY"] / med["SQ_WAVE_CYCLES"],
       "algorithmic_flops": flops, "achieved_TFLOPs": flops / statistics.median(dur) / 1e6}
json.dump(out, open(f"gpurun_out/r4/attn_prefill_pmc_b{B}.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters_median"}, indent=1))
PY
