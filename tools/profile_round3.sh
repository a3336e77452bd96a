#!/bin/bash
# Round-3 profiling session (run on the GPU box from the repo root): kernel trace of bench.py (headline leg
# only, then the configs[3] training step), then FETCH_SIZE / WRITE_SIZE passes (separate runs) over
# tools/pmc_round3.py, tools/pmc_round2.py and tools/pmc_scan.py.
# Every step under its own time limit; the script stops at the first failure.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "[profile] bench trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 $R/bench.py --workload train320 --steps 3 --warmup 2 > $O/train_under_rocprof.json 2> $O/train_under_rocprof.err
echo "[profile] train320 trace done"
for w in r3:pmc_round3.py r2:pmc_round2.py sc:pmc_scan.py; do
  t=${w%%:*}; s=${w##*:}
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${t}k -o ${t}k -- python3 $R/tools/$s > $O/${t}k.out 2>&1
  echo "[profile] $s kernel trace done"
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${t}f -o ${t}f -- python3 $R/tools/$s > $O/${t}f.out 2>&1
  echo "[profile] $s FETCH_SIZE pass done"
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${t}w -o ${t}w -- python3 $R/tools/$s > $O/${t}w.out 2>&1
  echo "[profile] $s WRITE_SIZE pass done"
done
find $O -name "*.csv" | xargs ls -la
