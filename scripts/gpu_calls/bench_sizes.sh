# Sanity + numbers for the other BASELINE configs (13B act-order, 33B g32 act-order, 65B): one protocol pass each.
mkdir -p gpurun_out/sizes
timeout 600 python bench.py --model 13b --act-order --steps 1 --warmup 1 --gen 64 --no-cpu-baseline --no-roofline-probe > gpurun_out/sizes/b13.json 2> gpurun_out/sizes/b13.err
timeout 900 python bench.py --model 33b --groupsize 32 --act-order --steps 1 --warmup 1 --gen 64 --no-cpu-baseline --no-roofline-probe > gpurun_out/sizes/b33.json 2> gpurun_out/sizes/b33.err
timeout 1200 python bench.py --model 65b --steps 1 --warmup 1 --gen 64 --no-cpu-baseline --no-roofline-probe > gpurun_out/sizes/b65.json 2> gpurun_out/sizes/b65.err
for f in b13 b33 b65; do python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/sizes/$f.json')); print('$f', d['prefill_tokens_per_s'], d['value'], d['decode_best_tokens_per_s'])
except Exception as e:
    print('$f FAILED', e); print(open('gpurun_out/sizes/$f.err').read()[-800:])
"; done
