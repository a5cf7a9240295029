#!/bin/bash
mkdir -p gpurun_out/ffn3
for rep in 1 2; do
for l in tools/ab/libA.so tinydiffusionmodels_amd/csrc/libtdm_hip.so tools/ab/libC.so; do
  echo "== $l M=32768"; TDM_HIP_LIB=$PWD/$l timeout -k 10 200 python tools/time_ffn.py --M 32768 --check-rows 256 2>&1 | grep -E "nprod 3|mismatch|Error|error"
  echo "== $l M=4096"; TDM_HIP_LIB=$PWD/$l timeout -k 10 200 python tools/time_ffn.py --M 4096 --check-rows 256 2>&1 | grep -E "nprod 3 mode|Error|error"
done; done > gpurun_out/ffn3/time.log 2>&1
bash tools/ab_text.sh tools/ab/libA.so > gpurun_out/ffn3/ab_AB.log 2>&1
bash tools/ab_text.sh tools/ab/libA.so tools/ab/libC.so > gpurun_out/ffn3/ab_AC.log 2>&1
cat gpurun_out/ffn3/*.log
