#!/bin/bash
mkdir -p gpurun_out/m1
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "marks or epoch" > gpurun_out/m1/t.log 2>&1; tail -3 gpurun_out/m1/t.log
timeout -k 10 600 python -m pytest tests/test_gpu_text.py -x -q -m gpu > gpurun_out/m1/tt.log 2>&1; tail -3 gpurun_out/m1/tt.log
timeout -k 10 500 python bench.py --sample-steps 0 --text-steps 0 --no-cpu-baseline > gpurun_out/m1/bench.json 2> gpurun_out/m1/bench.err; tail -c 600 gpurun_out/m1/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/m1/bench.json'))
r=d['roofline']
print(d['value'], d['steady_state']['steps_per_s'])
print({k:r[k] for k in ('ms_per_launch','achieved','frac')}, r.get('in_step'))
s=r.get('second',{})
print({k:s.get(k) for k in ('ms_per_launch','achieved','frac')}, s.get('in_step'))
P
