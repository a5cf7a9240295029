#!/bin/bash
# Same-box A/B of the text-denoiser train step with two builds of libtdm_hip.so:
#   tools/ab_text.sh tools/ab/libA.so [tools/ab/libB.so (default: the in-tree build)]
A=$1; B=${2:-tinydiffusionmodels_amd/csrc/libtdm_hip.so}
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    v=$(TDM_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --sample-steps 0 --text-steps 30 --no-cpu-baseline --no-launch-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['text_denoiser']; print(d['ms_per_step'], d['ms_per_step_dropout0'], d['other_gemm_mode']['ms_per_step'])")
    echo "$lib $v"
  done
done
