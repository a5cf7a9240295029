# Kernel traces of the UNet train step on one box: the default two-queue step (overlap table) and the one-queue step (timeline).
#   tools/prof_step.sh OUTDIR
O=${1:-gpurun_out/profstep}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/ov -- python bench.py --steps 50 --warmup 10 --sample-steps 0 --sample-chains 0 --text-steps 0 --no-cpu-baseline --no-launch-table --detail-out $O/ov.json > $O/ov.log 2>&1
python tools/step_overlap.py $(ls $O/ov/*/*kernel_trace.csv | head -1) > $O/step_overlap.txt; tail -2 $O/step_overlap.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/seq -- python bench.py --steps 50 --warmup 10 --sample-steps 0 --sample-chains 0 --text-steps 0 --no-cpu-baseline --no-launch-table --no-overlap --detail-out $O/seq.json > $O/seq.log 2>&1
python tools/step_timeline.py $(ls $O/seq/*/*kernel_trace.csv | head -1) > $O/step_timeline.txt; tail -2 $O/step_timeline.txt
rm -rf $O/ov $O/seq
