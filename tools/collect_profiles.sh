# Copy the newest outputs of tools/roundend2.sh (gpurun_out/r2, merged back from the GPU box) into profiles/ under the
# round's prefix:   bash tools/collect_profiles.sh r02
set -e
P=${1:-r02}; O=gpurun_out/r2
cp $O/bench.json profiles/${P}_bench.json
cp $O/bench_kernel_stats.csv profiles/${P}_bench_kernel_stats.csv
cp $O/bench_kernel_summary.txt profiles/${P}_bench_kernel_summary.txt
cp $O/step_timeline.txt profiles/${P}_step_timeline.txt
cp $O/kernel_pmc.json profiles/${P}_kernel_pmc.json
for c in FETCH_SIZE WRITE_SIZE TCC SQ; do cp $(ls -t $O/pmc_${c}_*_counter_collection.csv | head -1) profiles/${P}_pmc_${c}_counter_collection.csv; done
cp $(ls -t $O/pmc_trace_*_kernel_trace.csv | head -1) profiles/${P}_pmc_trace_kernel_trace.csv
ls -la profiles/${P}_*
