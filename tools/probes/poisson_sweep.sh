# Poisson sweep of DESIGN.md section 6b with the current library (96 requests, 576 + 128-token prompts, 128 generated,
# collocated EPD, prefill priority, 2048-token budget):  bash tools/probes/poisson_sweep.sh [rates...]
for r in ${@:-4 8 12 16 24 32}; do
  python tools/bench_engine.py --model 7b --requests 96 --rate $r --max-tokens 128 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); d=d[0] if isinstance(d,list) else d
print('| $r |', round(d['output_tok_s']), '|', round(d['ttft_p50_ms']), '/', round(d['ttft_p99_ms']), '|', d['tpot_p50_ms'], '/', d['tpot_p99_ms'], '|')"
done
