#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -o b1 -- python3 $R/tools/trace_b1.py > $O/b1.out 2>&1
echo "[profile] b1 trace done"
