#!/bin/bash
# kernel trace of the training step (configs[3]) for the per-kernel breakdown
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 $R/bench.py --workload train320 --steps 4 --warmup 2 > $O/train_under_rocprof.json 2> $O/train_under_rocprof.err
echo "[profile] train trace done"
