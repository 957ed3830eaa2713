# A/B of one environment switch over the whole decode step: fresh process per run, interleaved.
#   bash tools/probes/ab_env.sh HX_WIDE_SILU 0 1 [reps=4] [extra bench flags]
VAR=$1; A=$2; B=$3; REPS=${4:-4}; shift 4 || true
for rep in $(seq $REPS); do
for val in $A $B; do
  env $VAR=$val python bench.py --steps 20 --warmup 5 --no-ttft --no-cpu-baseline --no-serving --no-null-step "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); l=d.get('llava_13b') or {'ms_per_step':0,'whole_step_64':{'ms_per_step':0}}
print('$VAR[$val]', d['ms_per_step'], d['whole_step_64']['ms_per_step'], d['whole_step_64']['launches_per_layer'], l['ms_per_step'], l['whole_step_64']['ms_per_step'])"
done
done | tee /tmp/ab.txt
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    k = l.split(']')[0] + ']'
    d[k].append([float(x) for x in l.split(']')[1].split()])
print("medians: 7B 32 rows | 7B 64 rows | launches per layer | 13B 32 rows | 13B 64 rows (ms per step)")
for k, v in d.items():
    print(k, " | ".join("%.4f" % statistics.median(c) for c in zip(*v)))
PY
