#!/bin/bash
# Round-2 profiling session (run on the GPU box from the repo root): kernel trace of bench.py, then
# FETCH_SIZE / WRITE_SIZE passes (separate runs) over tools/pmc_round2.py and tools/pmc_scan.py.
# Every step under its own time limit; the script stops at the first failure.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "[profile] bench trace done"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2k -o r2k -- python3 $R/tools/pmc_round2.py all > $O/r2k.out 2>&1
echo "[profile] round-2 kernels trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r2f -o r2f -- python3 $R/tools/pmc_round2.py all > $O/r2f.out 2>&1
echo "[profile] FETCH_SIZE pass done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r2w -o r2w -- python3 $R/tools/pmc_round2.py all > $O/r2w.out 2>&1
echo "[profile] WRITE_SIZE pass done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/scf -o scf -- python3 $R/tools/pmc_scan.py > $O/scf.out 2>&1
echo "[profile] scan FETCH_SIZE pass done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/scw -o scw -- python3 $R/tools/pmc_scan.py > $O/scw.out 2>&1
echo "[profile] scan WRITE_SIZE pass done"
find $O -name "*.csv" | xargs ls -la
