# Round-5 evidence run on the GPU box (one gpurun call): tests, the default bench (short line + detail file), kernel trace of the
# bench, step timelines (two queues / one queue), PMC passes over single launches of the UNet step and over eager text steps
# (separate passes; rocprofv3 directly in front of python).  Outputs under gpurun_out/r5/ — copied to profiles/r05_* afterwards.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5
# launch ids of the default train step (tdm_unet_launch_name): 8 rb4.conv1 fwd (phase form), 16 rb4.conv1 dgrad up(h3) part (s2d),
# 15 its h1 part, 2 rb1.conv2 fwd, 12 rb4.conv2 dgrad, 6 rb3.conv1 fwd, 13 rb4.conv1 wgrad A, 11 rb4.conv2 wgrad, 18 rb3.conv2 wgrad, 10 out_bwd, 27 combine_dh1
IDS=8,16,15,2,12,6,13,11,18,10,27
rm -rf $O && mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -n 3 $O/pytest_gpu.log
timeout -k 10 700 python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python bench.py --steps 50 --warmup 10 --sample-steps 20 --sample-chains 0 --text-steps 5 --no-cpu-baseline --detail-out $O/prof_bench_detail.json > $O/prof_bench.log 2>&1
f=$(ls $O/prof_bench/*/*kernel_trace.csv | head -1)
python tools/prof_summary.py $f > $O/bench_kernel_summary.txt; head -12 $O/bench_kernel_summary.txt
python tools/step_overlap.py $f > $O/step_overlap.txt; tail -2 $O/step_overlap.txt       # the default step: two queues in the backward
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
# the same step on ONE queue (--no-overlap): every launch in order, durations without a co-running kernel
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_seq -- python bench.py --steps 50 --warmup 10 --sample-steps 0 --sample-chains 0 --text-steps 0 --no-cpu-baseline --no-launch-table --no-overlap --detail-out $O/prof_seq_detail.json > $O/prof_seq.log 2>&1
python tools/step_timeline.py $(ls $O/prof_seq/*/*kernel_trace.csv | head -1) > $O/step_timeline.txt; tail -2 $O/step_timeline.txt
rm -rf $O/prof_seq
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/pmc/trace -- python tools/pmc_replay.py --iters 21 --ids $IDS --names-out $O/launch_names.json > $O/pmc_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc/$c -- python tools/pmc_replay.py --iters 5 --ids $IDS > $O/pmc_$c.log 2>&1; done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc/SQ -- python tools/pmc_replay.py --iters 5 --ids $IDS > $O/pmc_SQ.log 2>&1
python tools/pmc_by_id.py $O/pmc $O/conv_traffic.json --ids $IDS --iters 5 --trace-iters 21 --names $O/launch_names.json --script tools/roundend5.sh
for d in $O/pmc/*; do for f in $(ls $d/*/*counter_collection.csv $d/*/*kernel_trace.csv 2>/dev/null); do python tools/keep_library_rows.py $f $O/pmc_$(basename $d)_$(basename $f); done; done
rm -rf $O/prof_bench $O/pmc
# text step: kernel trace + FETCH / WRITE PMC over eager steps at config 5's size, folded per kernel
P=$O/pmct; mkdir -p $P
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $P/trace -- python tools/text_steps.py 0.1 3 > $P/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $P/$c -- python tools/text_steps.py 0.1 3 > $P/$c.log 2>&1; done
python tools/pmc_text_fold.py $P $O/text_kernel_pmc.json > $O/text_kernel_pmc.txt; head -30 $O/text_kernel_pmc.txt
rm -rf $P
timeout -k 10 300 python tools/parity_report.py --out $O/parity.json > $O/parity.log 2>&1; tail -2 $O/parity.log
