#!/bin/bash
# Same-box A/B of two builds of libtdm_hip.so (boxes of the pool differ by up to 8 %):
#   tools/ab_bench.sh tools/ab/libA.so [tools/ab/libB.so (default: the in-tree build)] [extra bench.py flags]
A=$1; B=${2:-tinydiffusionmodels_amd/csrc/libtdm_hip.so}; shift; shift
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    v=$(TDM_HIP_LIB=$PWD/$lib python bench.py --steps 300 --warmup 20 --text-steps 0 --no-cpu-baseline --no-launch-table "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('sampling',{}).get('ms_per_reverse_step'))")
    echo "$lib $v"
  done
done
