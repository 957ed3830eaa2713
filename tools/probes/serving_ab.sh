# the serving legs with the decode loop's copies as kernels (1) or hipMemcpyAsync (0): fresh process per run, interleaved
for rep in 1 2 3; do
for kc in 0 1; do
  HX_ENGINE_KERNEL_COPIES=$kc python bench.py --steps 20 --warmup 5 --no-ttft --no-cpu-baseline --no-13b --no-null-step 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['serving']; t=s['twice_the_batch']
print('KC[$kc]', s['output_tok_s'], s['ttft_p50_ms'], s['tpot_p50_ms'], s['tpot_p99_ms'], '| 64:', t['output_tok_s'], t['ttft_p50_ms'], t['tpot_p50_ms'], t['tpot_p99_ms'])"
done
done
