#!/bin/bash
# Does the 13B leg run slower behind the 7B legs of a default line than on its own?  Kernel traces of both, cut into the
# 13B decode steps (tools/layer_timeline.py):  gpurun_out/$ROUND/13b_alone_timeline.md, 13b_after7b_timeline.md
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}
OUT=$R/gpurun_out/${ROUND:-r6}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
F="--steps 20 --no-cpu-baseline --no-serving --no-ttft --no-null-step --no-ragged"
rm -rf /tmp/p13a /tmp/p13b
rocprofv3 --kernel-trace --output-format csv -d /tmp/p13a -- python3 "$R/bench.py" $F --model 13b > "$OUT/13b_alone.json" 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d /tmp/p13b -- python3 "$R/bench.py" $F > "$OUT/13b_after7b.json" 2>/dev/null
python3 "$R/tools/layer_timeline.py" /tmp/p13a "$OUT/13b_alone_timeline.md" > /dev/null
N=$(sed -n 's/.* with \([0-9]*\) kernels.*/\1/p' "$OUT/13b_alone_timeline.md" | head -1)
TIMELINE_KERNELS=$N python3 "$R/tools/layer_timeline.py" /tmp/p13b "$OUT/13b_after7b_timeline.md" > /dev/null
head -14 "$OUT/13b_alone_timeline.md"; echo ----; head -14 "$OUT/13b_after7b_timeline.md"
