"""Probe: fp32 GEMM throughput of the mSRU projection shapes under the BLAS backends torch can
route to (rocBLAS / hipBLASLt / TunableOp).  Prints ms and TFLOP/s."""
import os
import sys
import time

import torch

dev = "cuda:0"


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


shapes = [("mSRU L1/L2  163840x1005 . 1005x3015", 163840, 1005, 3015),
          ("  padded K  163840x1008 . 1008x3015", 163840, 1008, 3015),
          ("  padded KN 163840x1008 . 1008x3072", 163840, 1008, 3072),
          ("  2/3 cols  163840x1005 . 1005x2010", 163840, 1005, 2010),
          ("fold A      8192x2696 . 2696x4020", 8192, 2696, 4020),
          ("filters     640x2000 . 2000x26960", 640, 2000, 26960)]


def run(tag):
    for name, M, K, N in shapes:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(K, N, device=dev)
        ms = timeit(lambda: torch.mm(a, b))
        print("%-10s %-40s %.3f ms  %.1f TF" % (tag, name, ms, 2.0 * M * K * N / ms / 1e9), flush=True)
        del a, b


mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode in ("cublas", "hipblaslt", "cublaslt"):
    torch.backends.cuda.preferred_blas_library(mode)
if mode == "tunable":
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.set_filename(os.path.join("gpurun_out", "tunableop.csv"))
    torch.cuda.tunable.set_max_tuning_duration(2000)
print("mode", mode, "preferred", torch.backends.cuda.preferred_blas_library(), flush=True)
run(mode)
