cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pt2; rm -rf $O; mkdir -p $O
for p in 0.1 0.0; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/p$p -- python tools/text_steps.py $p 8 > $O/log$p.txt 2>&1
f=$(ls $O/p$p/*/*kernel_trace.csv | head -1)
python tools/prof_summary.py $f > $O/summary_$p.txt
rm -rf $O/p$p
done
