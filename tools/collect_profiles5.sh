# Copy the outputs of tools/roundend5.sh (gpurun_out/r5, merged back from the GPU box) into profiles/ as r05_*
set -e
O=gpurun_out/r5; P=profiles/r05
cp $O/bench.json ${P}_bench.json
cp $O/bench_detail.json ${P}_bench_detail.json
cp $O/bench_kernel_stats.csv ${P}_bench_kernel_stats.csv
cp $O/bench_kernel_summary.txt ${P}_bench_kernel_summary.txt
cp $O/step_timeline.txt ${P}_step_timeline.txt
cp $O/step_overlap.txt ${P}_step_overlap.txt
cp $O/conv_traffic.json ${P}_conv_traffic.json
cp $O/launch_names.json ${P}_launch_names.json
cp $O/parity.json ${P}_parity.json
cp $O/text_kernel_pmc.json ${P}_text_kernel_pmc.json
cp $O/text_kernel_pmc.txt ${P}_text_kernel_pmc.txt
for c in FETCH_SIZE WRITE_SIZE SQ; do cp $(ls -t $O/pmc_${c}_*_counter_collection.csv | head -1) ${P}_pmc_${c}_counter_collection.csv; done
cp $(ls -t $O/pmc_trace_*_kernel_trace.csv | head -1) ${P}_pmc_trace_kernel_trace.csv
tail -n 3 $O/pytest_gpu.log > ${P}_pytest_gpu_tail.txt
ls -la ${P}_* | awk '{print $5, $9}'
