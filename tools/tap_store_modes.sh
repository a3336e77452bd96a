#!/bin/bash
# A/B of the tap tile kernel's load policy (DMN_TAP_LOAD = 0 all non-temporal, 1 all plain, 2 plain for the rows a
# neighbouring band re-reads as halo): parity tests, time, and FETCH_SIZE per launch.  (The same script compared
# the STORE policies earlier in round 2: profiles/r2_tap_store_modes.txt.)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/tapmodes
mkdir -p $O
export PYTHONPATH=$R
for m in 0 1 2; do
  export DMN_TAP_LOAD=$m
  cd $R
  timeout -k 10 120 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tap" --timeout 100 --timeout-method thread > $O/pytest_$m.log 2>&1
  timeout -k 10 60 python -u tools/tap_bench.py > $O/bench_$m.log 2>&1
  cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/w$m -o w -- python3 $R/tools/pmc_round2.py tap > $O/w$m.out 2>&1
  echo "[tapmodes] mode $m done"
done
