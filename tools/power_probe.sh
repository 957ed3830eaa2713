#!/bin/bash
# Clocks and power of the GPU while a decode loop runs (rocm-smi polled every ~0.3 s from start to end): is the step power-limited?
#   tools/power_probe.sh <tag> [ab_executors.py EXECS value]
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
TAG=$1
OUT=$R/gpurun_out/r4
mkdir -p "$OUT"
EXECS="${2:-plan}" python3 "$R/tools/ab_executors.py" 7b 200 20 > "$OUT/${TAG}_ab.txt" 2>&1 &
PID=$!
: > "$OUT/${TAG}_smi.txt"
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power \(W\)|Sensor (junction|memory)" | sed 's/.*: //' | tr '\n' ' ' >> "$OUT/${TAG}_smi.txt"
  echo >> "$OUT/${TAG}_smi.txt"
  sleep 0.3
done
tail -2 "$OUT/${TAG}_ab.txt"
sort "$OUT/${TAG}_smi.txt" | uniq -c | sort -rn | head -12
