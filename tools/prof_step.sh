#!/bin/bash
# Kernel trace of the bench command cut into decode steps / one middle layer (tools/layer_timeline.py).
#   tools/prof_step.sh <tag> [extra bench.py args]   ->  gpurun_out/$ROUND/<tag>_timeline.md (ROUND defaults to r6), <tag>_kernel_stats.csv, <tag>_bench.json
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}
TAG=$1; shift
OUT=$R/gpurun_out/${ROUND:-r6}
mkdir -p "$OUT"
D=/tmp/prof_$TAG
rm -rf "$D"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 "$R/bench.py" --steps 64 --warmup 3 \
    --no-cpu-baseline --no-serving --no-ttft --no-13b "$@" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err" || { tail -5 "$OUT/${TAG}_bench.err"; exit 1; }
python3 "$R/tools/layer_timeline.py" "$D" "$OUT/${TAG}_timeline.md" "$OUT/${TAG}_in_step.json" > /dev/null
cp "$(find "$D" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
cat "$OUT/${TAG}_timeline.md"
