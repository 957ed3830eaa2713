# does the pool's plane skew change the serving leg?  (TTFT p50 of 32 requests at t = 0 went 268 -> 421 ms in one run)
for skew in 0 768 0 768; do
  HX_KV_POOL_SKEW=$skew python bench.py --steps 20 --warmup 5 --no-ttft --no-cpu-baseline --no-13b --no-null-step 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['serving']; t=s['twice_the_batch']
print('SKEW[$skew]', d['ms_per_step'], 'serving', s['output_tok_s'], s['ttft_p50_ms'], s['tpot_p50_ms'], '| 64:', t['output_tok_s'], t['ttft_p50_ms'], t['tpot_p50_ms'])"
done
