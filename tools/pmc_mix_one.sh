#!/bin/bash
# rocprofv3 passes over the one-pass mix backward (tools/mix_bwd_bench.py, one-pass mode only): kernel stats, SQ
# wave-cycle breakdown, HBM fetch / write sizes.  Run from the repo root on the GPU box; results in gpurun_out/pmc_mix_one/.
export TMPDIR=/tmp MIXBENCH_MODES=1 MIXBENCH_REPS=3
out=$PWD/gpurun_out/pmc_mix_one
mkdir -p $out
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $out/stats -o s -- python3 tools/mix_bwd_bench.py > $out/stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $out/sq -o s -- python3 tools/mix_bwd_bench.py > $out/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o s -- python3 tools/mix_bwd_bench.py > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/write -o s -- python3 tools/mix_bwd_bench.py > $out/write.log 2>&1 || exit 1
find $out -name "*.csv" | head -20
