#!/bin/bash
# kernel trace of bench.py cut into the 64-row steps (whole_step_64):  bash tools/probes/prof64.sh <tag> [bench flags; default --no-serving]
set -eu
R=${GRAFT_REPO_ROOT:?}
TAG=${1:-w64}; shift || true
FLAGS=${*:---no-serving}
OUT=$R/gpurun_out/r5; mkdir -p "$OUT"
D=/tmp/prof_$TAG; rm -rf "$D"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$D" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline $FLAGS --no-ttft --no-13b --no-null-step > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err" || { tail -5 "$OUT/${TAG}_bench.err"; exit 1; }
TIMELINE_MUST_CONTAIN=gemm_xreg_wide python3 "$R/tools/layer_timeline.py" "$D" "$OUT/${TAG}_timeline.md" > /dev/null
cat "$OUT/${TAG}_timeline.md"
