# kernel-trace of six eager full text train steps at B = 256 under TDM_TN_RING = 0 / 1 (same box): per-kernel totals side by side
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pfab; rm -rf $O; mkdir -p $O
for x in 0 1; do
  export TDM_TN_RING=$x
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/p$x -- python tools/text_full_steps.py 256 6 0 > $O/log$x.txt 2>&1
  f=$(ls $O/p$x/*/*kernel_trace.csv | head -1)
  python tools/prof_summary.py $f > $O/summary$x.txt
  rm -rf $O/p$x
done
unset TDM_TN_RING
