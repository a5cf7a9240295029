# Kernel traces of the default UNet bench (two queues in the backward) and of the one-queue step: step_overlap / step_timeline / kernel summary
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python bench.py --steps 50 --warmup 10 --sample-steps 20 --sample-chains 0 --text-steps 5 --no-cpu-baseline --detail-out $O/prof_bench_detail.json > $O/prof_bench.log 2>&1
f=$(ls $O/prof_bench/*/*kernel_trace.csv | head -1)
python tools/prof_summary.py $f > $O/bench_kernel_summary.txt; head -12 $O/bench_kernel_summary.txt
python tools/step_overlap.py $f > $O/step_overlap.txt; tail -2 $O/step_overlap.txt
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_seq -- python bench.py --steps 50 --warmup 10 --sample-steps 0 --sample-chains 0 --text-steps 0 --no-cpu-baseline --no-launch-table --no-overlap --detail-out $O/prof_seq_detail.json > $O/prof_seq.log 2>&1
python tools/step_timeline.py $(ls $O/prof_seq/*/*kernel_trace.csv | head -1) > $O/step_timeline.txt; tail -2 $O/step_timeline.txt
rm -rf $O/prof_seq $O/prof_bench
