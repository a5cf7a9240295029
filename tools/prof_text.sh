cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pt; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python bench.py --steps 2 --warmup 1 --sample-steps 0 --text-steps 10 --no-cpu-baseline --no-launch-table > $O/log.txt 2>&1
f=$(ls $O/p/*/*kernel_trace.csv | head -1)
python tools/prof_summary.py $f > $O/summary.txt
python tools/keep_library_rows.py $f $O/trace.csv
rm -rf $O/p
