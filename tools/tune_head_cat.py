"""TunableOp search for the head's single-operand level GEMMs Z[b] = Wt (9*Cout x (Cx+Cf)) . [x | feat][b]
(the backbone writes each tap behind room for x, dms_amd DMN._link_head_buffers).  tools/tune_gemms.py
tunes on the per-block merge path, which produces compact taps and therefore the two-GEMM head, so
these shapes are searched here.  Appends nothing by itself: prints / writes a table to merge."""
import os
import sys

import torch
import torch.cuda.tunable as tunable

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_head_cat.csv"
dev = "cuda:0"
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(40)
tunable.set_max_tuning_iterations(30)
tunable.set_filename(out)
for B in (32, 1):
    for hw, cout, k in ((256, 512, 1005 + 2688), (1024, 256, 512 + 1552), (4096, 128, 256 + 704), (16384, 64, 128 + 336)):
        a = torch.randn(9 * cout, k, device=dev) * 0.05
        x = torch.randn(B, k, hw, device=dev)
        torch.matmul(a, x)
        torch.cuda.synchronize()
        print("tuned B=%d" % B, hw, 9 * cout, k, flush=True)
for r in tunable.get_results():
    print(r)
