# HBM traffic of the prefill attention kernel (FETCH_SIZE / WRITE_SIZE in separate passes, as tools/pmc_decode.sh):
#   HX_PREFILL_B=32 bash tools/pmc_prefill_traffic.sh   ->  gpurun_out/r4/attn_prefill_traffic_b$B.json
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repo root (gpurun exports it)}
cd /tmp; export TMPDIR=/tmp
export HX_PREFILL_B=${HX_PREFILL_B:-4}
O=$R/gpurun_out/r4
mkdir -p "$O"
rm -rf "$O"/pmc_pft_*
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_pft_$c -o p -- python3 $R/tools/prof_attn_prefill32.py > $O/pmc_pft_$c.log 2>&1 \
    || { echo "PMC pass $c failed:" >&2; tail -5 $O/pmc_pft_$c.log >&2; exit 1; }
done
cd $R
python3 - <<'PY'
import csv, glob, statistics, json, os
B = int(os.environ.get("HX_PREFILL_B", "4"))
out = {"sequences_of_704_tokens": B}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/r4/pmc_pft_{c}/**/*counter_collection.csv", recursive=True)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "attn_fwd32" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c + "_KiB_median"] = statistics.median(v)
# gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md): doubled
out["hbm_read_MB"] = out["FETCH_SIZE_KiB_median"] * 1024 * 2 / 1e6
out["hbm_write_MB"] = out["WRITE_SIZE_KiB_median"] * 1024 / 1e6
H, D, n = 32, 128, 704
out["unique_bytes_MB"] = {"q": B * n * H * D * 2 / 1e6, "k_v": 2 * B * n * H * D * 2 / 1e6, "o": B * n * H * D * 2 / 1e6}
steps = 1 + 3 + 5 + 7 + 9 + 11
out["lds_dma_MB_requested"] = B * H * steps * 32768 / 1e6
json.dump(out, open(f"gpurun_out/r4/attn_prefill_traffic_b{B}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
