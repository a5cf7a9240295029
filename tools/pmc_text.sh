# PMC passes (FETCH_SIZE / WRITE_SIZE, separate) + kernel trace over three eager text-denoiser train steps at config 5's size;
# folded per kernel name + grid by tools/pmc_text_fold.py into gpurun_out/pmct/text_kernel_pmc.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmct; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python tools/text_steps.py 0.1 3 > $O/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python tools/text_steps.py 0.1 3 > $O/$c.log 2>&1; done
python tools/pmc_text_fold.py $O $O/text_kernel_pmc.json
rm -rf $O/trace $O/FETCH_SIZE $O/WRITE_SIZE
