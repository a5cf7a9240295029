# tools/sweep_env.sh OUTDIR VAR v1 v2 ...: the UNet train step under each value of an environment variable, two rounds, one box
O=$1; V=$2; shift 2; mkdir -p $O
for rep in 1 2; do for v in "$@"; do
env $V=$v timeout -k 10 200 python bench.py --steps 300 --warmup 30 --sample-steps 0 --sample-chains 0 --text-steps 0 --no-cpu-baseline --detail-out $O/b_${v}_$rep.json > $O/l_${v}_$rep.json 2> $O/e_${v}_$rep.err
python - "$O/b_${v}_$rep.json" "$V=$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); lt = d["launch_table"]["all_us"]
print(f"{sys.argv[2]}: steps/s {d['value']:.1f} steady {d['steady_state']['steps_per_s']}  group_sums (30) {lt.get('30')} first_wgrad (31) {lt.get('31')} reduce (32) {lt.get('32')}")
PY
done; done
