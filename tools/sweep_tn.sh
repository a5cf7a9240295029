for w in 256 384 512 768; do
  v=$(TDM_TN_WGS=$w python bench.py --steps 5 --warmup 2 --sample-steps 0 --text-steps 30 --no-cpu-baseline --no-launch-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['text_denoiser']; print(d['ms_per_step'], d['ms_per_step_dropout0'], d['other_gemm_mode']['ms_per_step'])")
  echo "TDM_TN_WGS=$w $v"
done
