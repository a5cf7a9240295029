#!/bin/bash
# Same-box A/B of the text-denoiser train step with two builds of libtdm_hip.so:
#   tools/ab_text.sh tools/ab/libA.so [tools/ab/libB.so (default: the in-tree build)]
A=$1; B=${2:-tinydiffusionmodels_amd/csrc/libtdm_hip.so}
O=gpurun_out/abtextlib; mkdir -p $O
for rep in 1 2 3; do
  i=0
  for lib in "$A" "$B"; do
    i=$((i+1))
    TDM_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 5 --warmup 2 --sample-steps 0 --sample-chains 0 --text-steps 30 --no-cpu-baseline --no-launch-table --detail-out $O/b_${i}_$rep.json > $O/line_${i}_$rep.json 2> $O/b_${i}_$rep.err
    python - "$O/b_${i}_$rep.json" "$lib" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = d.get("text_denoiser", {})
print(f"{sys.argv[2]}: denoiser ms/step {t.get('ms_per_step')}  dropout0 {t.get('ms_per_step_dropout0')}  other gemm mode {t.get('other_gemm_mode', {}).get('ms_per_step')}")
PY
  done
done
