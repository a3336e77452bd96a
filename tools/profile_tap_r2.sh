#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / kernel-trace passes over the tap_gather levels (final store / load policy)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tapf -o tapf -- python3 $R/tools/pmc_round2.py tap > $O/tapf.out 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/tapw -o tapw -- python3 $R/tools/pmc_round2.py tap > $O/tapw.out 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tapk -o tapk -- python3 $R/tools/pmc_round2.py tap > $O/tapk.out 2>&1
echo "[profile] tap passes done"
