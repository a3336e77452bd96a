#!/bin/bash
# rocprofv3 counter passes over dmn_dynfilter_fwd at B = 32, 512 px (matrix-core kernel, LDS-DMA ring kernel).
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_dynfilter
mkdir -p $out
cat > /tmp/df_once.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dms_amd import ops
os.environ["DMN_DYNFILTER_DMA"] = "1"
B, L, K, C, S = 32, 20, 10, 2696, 256
g = torch.Generator(device="cuda").manual_seed(1)
V = torch.randn((B, C, S), device="cuda", generator=g); F = torch.rand((L, B, K, C), device="cuda", generator=g)
for _ in range(4):
    ops.dynfilter(V, F)
torch.cuda.synchronize()
PY
timeout -s KILL 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY --output-format csv -d $out/sq -o s -- python3 /tmp/df_once.py > $out/sq.log 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/tcc -o s -- python3 /tmp/df_once.py > $out/tcc.log 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq2 -o s -- python3 /tmp/df_once.py > $out/sq2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for d in ("sq", "tcc", "sq2"):
    for f in glob.glob("gpurun_out/pmc_dynfilter/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "dynfilter" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in sorted(acc): print(d, k, acc[k] / max(n[k], 1))
PY
