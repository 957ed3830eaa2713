# A/B of one hx_debug_set_option over the whole decode step: fresh process per run, interleaved.
#   bash tools/probes/ab_option.sh "decode_seq_major=1" [reps=4] [extra bench flags]
OPT=$1; REPS=${2:-4}; shift; shift || true
for rep in $(seq $REPS); do
for which in off on; do
  if [ $which = on ]; then export HX_DEBUG_OPTIONS=$OPT; else unset HX_DEBUG_OPTIONS; fi
  python bench.py --steps 20 --warmup 5 --no-ttft --no-cpu-baseline --no-serving --no-null-step "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); l=d.get('llava_13b') or {'ms_per_step':0,'whole_step_64':{'ms_per_step':0}}
print('OPT[$which]', d['ms_per_step'], d['whole_step_64']['ms_per_step'], l['ms_per_step'], l['whole_step_64']['ms_per_step'], d['roofline']['avg_launch_us'])"
done
done | tee /tmp/ab.txt
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    k = l.split(']')[0] + ']'
    d[k].append([float(x) for x in l.split(']')[1].split()])
print("medians: 7B 32 rows | 7B 64 rows | 13B 32 rows | 13B 64 rows (ms per step) | standalone attention us")
for k, v in d.items():
    print(k, " | ".join("%.4f" % statistics.median(c) for c in zip(*v)))
PY
