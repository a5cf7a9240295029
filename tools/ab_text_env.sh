# Same-box A/B of the text-denoiser train step (config 5) and the full text step under an environment variable:
#   tools/ab_text_env.sh OUTDIR VAR VALUE_A VALUE_B
O=${1:-gpurun_out/abtext}; V=$2; A=$3; B=$4
mkdir -p $O
for rep in 1 2; do
  for x in $A $B; do
    env $V=$x timeout -k 10 300 python bench.py --steps 5 --warmup 2 --sample-steps 0 --sample-chains 0 --text-steps 30 --no-cpu-baseline --no-launch-table --detail-out $O/b_${x}_$rep.json > $O/line_${x}_$rep.json 2> $O/b_${x}_$rep.err
    python - "$O/b_${x}_$rep.json" "$V=$x" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
t = d.get("text_denoiser", {}); f = d.get("text_train_full", {})
print(f"{sys.argv[2]}: denoiser ms/step {t.get('ms_per_step')}  dropout0 {t.get('ms_per_step_dropout0')}  other gemm mode {t.get('other_gemm_mode', {}).get('ms_per_step')}  full text step {json.dumps(f)[:200]}")
PY
  done
done
