#!/bin/bash
# Copy the judged summaries of a tools/profile_round3.sh session from gpurun_out/prof_r3 into profiles/
# (run here, after gpurun merged the session's output back).
set -e
O=gpurun_out/prof_r3
cp $O/bench/bench_kernel_stats.csv profiles/r3_bench_kernel_stats.csv
python tools/trace_steps.py $O/bench/bench_kernel_trace.csv 40 > profiles/r3_bench_steady_state_per_step.txt
cp $O/bench_under_rocprof.json profiles/r3_bench_under_rocprof.json
head -60 $O/train/train_kernel_stats.csv > profiles/r3_train320_kernel_stats_top60.csv
cp $O/train_under_rocprof.json profiles/r3_train320_under_rocprof.json
python tools/trace_steps.py $O/train/train_kernel_trace.csv 60 | cut -c1-200 > profiles/r3_train320_steady_state_per_step.txt
cp $O/r3k/r3k_kernel_stats.csv profiles/r3_kernels_kernel_stats.csv
cp $O/sck/sck_kernel_stats.csv profiles/r3_scan_kernel_stats.csv
for p in r3 r2 sc; do
  for t in f:fetch w:write; do
    k=${t%%:*}; n=${t##*:}
    (head -1 $O/$p$k/$p${k}_counter_collection.csv; grep "dmn::" $O/$p$k/$p${k}_counter_collection.csv) > profiles/r3_pmc_${p}_${n}_size.csv
  done
done
python tools/pmc_summary_r3.py $O > profiles/traffic_r3.json
ls -la profiles/r3_*
