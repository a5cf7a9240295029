O=gpurun_out/r5o; mkdir -p $O
for rep in 1 2; do for v in 256 128 96 64; do
TDM_NS28=$v timeout -k 10 200 python bench.py --steps 300 --warmup 30 --sample-steps 0 --sample-chains 0 --text-steps 0 --no-cpu-baseline --detail-out $O/b_${v}_$rep.json > $O/l_${v}_$rep.json 2> $O/e_${v}_$rep.err
python - "$O/b_${v}_$rep.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); lt = d["launch_table"]["all_us"]
print(f"NS28={sys.argv[2]}: steps/s {d['value']:.1f} steady {d['steady_state']['steps_per_s']}  wg rb4c2 (11) {lt['11']} s2d (13) {lt['13']} rb4c1B (14) {lt['14']} rb1c2 (28) {lt['28']} first_wgrad (32) {lt.get('32')} reduce (33) {lt.get('33')}")
PY
done; done
