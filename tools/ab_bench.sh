#!/bin/bash
# Same-box A/B of two builds of libtdm_hip.so (boxes of the pool differ by up to 8 %):
#   tools/ab_bench.sh tools/ab/libA.so [tools/ab/libB.so (default: the in-tree build)]
A=$1; B=${2:-tinydiffusionmodels_amd/csrc/libtdm_hip.so}
O=gpurun_out/abbench; mkdir -p $O
for rep in 1 2 3; do
  i=0
  for lib in "$A" "$B"; do
    i=$((i+1))
    TDM_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 30 --sample-steps 100 --sample-chains 0 --text-steps 0 --no-cpu-baseline --detail-out $O/b_${i}_$rep.json > $O/line_${i}_$rep.json 2> $O/b_${i}_$rep.err
    python - "$O/b_${i}_$rep.json" "$lib" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); lt = d.get("launch_table", {}).get("all_us", {})
print(f"{sys.argv[2]}: steps/s {d['value']:.1f}  ms/step {d['ms_per_step']:.4f}  sum alone {d.get('launch_table', {}).get('sum_us')}  14x14 64->64 fwd (ids 5,6,7) {lt.get('5')} {lt.get('6')} {lt.get('7')}  dgrads (19,21,24) {lt.get('19')} {lt.get('21')} {lt.get('24')}  28x28 (2,12,29) {lt.get('2')} {lt.get('12')} {lt.get('29')}  sampling ms/rev {d.get('sampling', {}).get('ms_per_reverse_step')}")
PY
  done
done
