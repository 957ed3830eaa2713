set -u
export ROUND=r6
bash tools/prof_step.sh r6_bench7b --no-null-step --no-ragged > /dev/null 2>&1 || echo "prof_step 7b failed"
bash tools/prof_step.sh r6_bench13b --model 13b --no-null-step --no-ragged > /dev/null 2>&1 || echo "prof_step 13b failed"
bash tools/pmc_decode.sh > gpurun_out/r6/pmc.log 2>&1 || echo "pmc failed"
python bench.py --steps 20 > gpurun_out/r6/r6_bench7b_13b_steps20.json 2> gpurun_out/r6/r6_bench7b_13b_steps20.err || echo "bench failed"
ls -la gpurun_out/r6/
