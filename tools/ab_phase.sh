# A/B of rb4.conv1's phase form on ONE box: TDM_RB4_PHASE=0 (nine taps over the up-sampled image) vs the default.
#   tools/ab_phase.sh OUTDIR
O=${1:-gpurun_out/abph}
mkdir -p $O
for rep in 1 2; do
  for ph in 0 1; do
    TDM_RB4_PHASE=$ph timeout -k 10 200 python bench.py --steps 300 --warmup 30 --sample-steps 100 --sample-chains 0 --text-steps 0 --no-cpu-baseline --detail-out $O/b_${ph}_$rep.json > $O/line_${ph}_$rep.json 2> $O/b_${ph}_$rep.err
    python - "$O/b_${ph}_$rep.json" "$ph" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
lt = d.get("launch_table", {}).get("all_us", {})
print(f"phase={sys.argv[2]} steps/s {d['value']:.1f}  ms/step {d['ms_per_step']:.4f}  rb4.conv1 fwd (id 8) {lt.get('8')} us  dgrad h1 (15) {lt.get('15')} + h3 (16) {lt.get('16')} us  (17) {lt.get('17')} us  sampling ms/rev {d.get('sampling', {}).get('ms_per_reverse_step')}")
PY
  done
done
