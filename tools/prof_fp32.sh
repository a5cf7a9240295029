O=gpurun_out/r5k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/fp32 -- python bench.py --conv-mode 0 --steps 30 --warmup 5 --sample-steps 0 --sample-chains 0 --text-steps 0 --no-cpu-baseline --no-launch-table --no-overlap --detail-out $O/fp32.json > $O/fp32.log 2>&1
python tools/step_timeline.py $(ls $O/fp32/*/*kernel_trace.csv | head -1) > $O/fp32_timeline.txt; tail -3 $O/fp32_timeline.txt
rm -rf $O/fp32
