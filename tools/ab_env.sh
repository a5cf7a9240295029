# Same-box A/B of the UNet train step under an environment variable:  tools/ab_env.sh OUTDIR VAR VALUE_A VALUE_B
O=${1:-gpurun_out/abenv}; V=$2; A=$3; B=$4
mkdir -p $O
for rep in 1 2; do
  for x in $A $B; do
    env $V=$x timeout -k 10 200 python bench.py --steps 300 --warmup 30 --sample-steps 0 --sample-chains 0 --text-steps 0 --no-cpu-baseline --detail-out $O/b_${x}_$rep.json > $O/line_${x}_$rep.json 2> $O/b_${x}_$rep.err
    python - "$O/b_${x}_$rep.json" "$V=$x" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
lt = d.get("launch_table", {})
print(f"{sys.argv[2]}: steps/s {d['value']:.1f}  ms/step {d['ms_per_step']:.4f}  steady {d['steady_state']['steps_per_s']}  sum of launches alone {lt.get('sum_us')} us  id13 {lt.get('all_us', {}).get('13')} us")
PY
  done
done
