"""TunableOp search for the beta = 1 GEMMs of the in-place DPN stages (last 1x1 conv of a block
accumulating onto [res | slot]).  The model runs them on row-slice views, which TunableOp's tuning
mode cannot duplicate, so they are tuned here on contiguous operands of the same m, n, k, ld (the
table key holds neither beta nor the batch stride).  Prints the new table lines."""
import os
import sys

os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
import torch
import torch.cuda.tunable as tunable

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_beta1.csv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = "cuda:0"
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(40)
tunable.set_max_tuning_iterations(30)
tunable.set_filename(out)
for hw, m, k in ((16384, 272, 96), (4096, 544, 192), (1024, 1048, 384), (256, 2176, 768)):
    a = torch.randn(m, k, device=dev) * 0.05
    x = torch.randn(B, k, hw, device=dev)
    acc = torch.randn(B, m, hw, device=dev)
    torch.baddbmm(acc, a.unsqueeze(0).expand(B, -1, -1), x, out=acc)
    torch.cuda.synchronize()
    print("tuned", hw, m, k, flush=True)
for r in tunable.get_results():
    print(r)
