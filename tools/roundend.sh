set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/pytest_gpu.log 2>&1; tail -n 3 gpurun_out/r2/pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err; tail -c 600 gpurun_out/r2/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_bench -- python bench.py --steps 50 --warmup 10 --sample-steps 20 --text-steps 5 --no-cpu-baseline > gpurun_out/r2/prof_bench.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/conv_alone -- python tools/pmc_conv.py 20 > gpurun_out/r2/conv_alone.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r2/pmc_$c -- python tools/pmc_conv.py 5 > gpurun_out/r2/pmc_$c.log 2>&1; done
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r2/pmc_TCC -- python tools/pmc_conv.py 5 > gpurun_out/r2/pmc_TCC.log 2>&1
for d in FETCH_SIZE WRITE_SIZE TCC; do python tools/pmc_summary.py gpurun_out/r2/pmc_$d; done
python tools/prof_summary.py $(ls gpurun_out/r2/conv_alone/*/*kernel_trace.csv | head -1) | head -5
