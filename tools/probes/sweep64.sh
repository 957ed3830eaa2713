# KV pool placement (hydrainfer_amd/memory/kv_pool.py): bytes between the (layer, k/v) planes of the pool, in the whole
# decode step; fresh process per run (the physical pages behind the pool differ from process to process), interleaved
# repetitions.    bash tools/probes/sweep64.sh [reps=4] [extra bench flags]
REPS=${1:-4}; shift || true
for rep in $(seq $REPS); do
for skew in 0 256 768 4352; do
  HX_KV_POOL_SKEW=$skew python bench.py --steps 20 --warmup 5 --no-ttft --no-cpu-baseline --no-serving --no-null-step "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); l=d.get('llava_13b') or {'ms_per_step':0,'whole_step_64':{'ms_per_step':0}}
print('SKEW[$skew]', d['ms_per_step'], d['whole_step_64']['ms_per_step'], l['ms_per_step'], l['whole_step_64']['ms_per_step'])"
done
done | tee /tmp/sweep.txt
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('/tmp/sweep.txt'):
    k = l.split(']')[0] + ']'
    d[k].append([float(x) for x in l.split(']')[1].split()])
print("medians: 7B 32 rows | 7B 64 rows | 13B 32 rows | 13B 64 rows (ms per step)")
for k, v in d.items():
    print(k, " | ".join("%.4f" % statistics.median(c) for c in zip(*v)))
PY
