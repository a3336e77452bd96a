#!/bin/bash
# Copy the judged summaries of a tools/profile_round2.sh session from gpurun_out/prof_r2 into profiles/
# (run here, after gpurun merged the session's output back).
set -e
O=gpurun_out/prof_r2
cp $O/bench/bench_kernel_stats.csv profiles/r2_bench_kernel_stats.csv
python tools/trace_steps.py $O/bench/bench_kernel_trace.csv 40 > profiles/r2_bench_steady_state_per_step.txt
cp $O/bench_under_rocprof.json profiles/r2_bench_under_rocprof.json
cp $O/r2k/r2k_kernel_stats.csv profiles/r2_kernels_kernel_stats.csv
for t in f:fetch w:write; do
  k=${t%%:*}; n=${t##*:}
  (head -1 $O/r2$k/r2${k}_counter_collection.csv; grep "dmn::" $O/r2$k/r2${k}_counter_collection.csv) > profiles/r2_pmc_${n}_size.csv
  (head -1 $O/sc$k/sc${k}_counter_collection.csv; grep "dmn::" $O/sc$k/sc${k}_counter_collection.csv) > profiles/r2_pmc_scan_${n}_size.csv
done
python tools/pmc_summary_r2.py $O > profiles/traffic_r2.json
ls -la profiles/r2_bench_kernel_stats.csv profiles/r2_pmc_fetch_size.csv profiles/traffic_r2.json
