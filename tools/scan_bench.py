#!/usr/bin/env python
"""Microbench of the SRU scan entry points over shapes / prefetch depths / chunk counts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops

dev = "cuda"


def run(T, B, d, k, h_t0, nch, depth, iters=10, want_c=False):
    if depth:
        os.environ["DMN_SRU_DEPTH"] = str(depth)
    else:
        os.environ.pop("DMN_SRU_DEPTH", None)
    g = torch.Generator(device=dev).manual_seed(1)
    u = torch.randn((T, B, d, k), device=dev, generator=g)
    x = torch.randn((T, B, d), device=dev, generator=g) if k == 3 else None
    bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g)
    for _ in range(2):
        ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, h_t0, want_c=want_c, nchunks=nch)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, h_t0, want_c=want_c, nchunks=nch)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    nbytes = (4 * k + (4 if k == 3 else 0)) * T * B * d + 4 * (T - h_t0) * B * d + (4 * T * B * d if want_c else 0)
    print(("+c " if want_c else "") + "T=%5d B=%3d d=%4d k=%d h_t0=%5d nch=%2d depth=%2s : %8.3f ms  %7.1f GB/s (%.1f%% of 8 TB/s)"
          % (T, B, d, k, h_t0, nch, depth or "-", ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80), flush=True)


def run_state(T, B, d, depth, iters=10):
    os.environ["DMN_SRU_DEPTH"] = str(depth)
    g = torch.Generator(device=dev).manual_seed(1)
    u2 = torch.randn((T, B, d, 2), device=dev, generator=g)
    bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g)
    for _ in range(2):
        ops.sru_state(u2, bias, None, nchunks=1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        ops.sru_state(u2, bias, None, nchunks=1)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    nbytes = 8 * T * B * d
    print("state-only T=%5d B=%3d d=%4d depth=%2s : %8.3f ms  %7.1f GB/s (%.1f%% of 8 TB/s)"
          % (T, B, d, depth, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "insitu"):
        for k in (4, 3):
            for nch, depth in [(1, 8), (1, 16), (1, 24), (2, 8), (2, 16), (4, 8), (4, 16), (8, 8)]:
                run(5120, 32, 1005, k, 0, nch, depth)
        run(5120, 32, 1005, 3, 5120 - 256, 1, 24)
    if which in ("all", "stress"):
        for depth in (4, 8, 16, 24):
            run(4096, 128, 1000, 3, 0, 1, depth)
        run(4096, 128, 1005, 4, 0, 1, 8)
        run(4096, 128, 1005, 4, 0, 1, 16)
    if which == "deep":
        for k in (4, 3):
            for depth in (20, 24, 32, 40):
                run(5120, 32, 1005, k, 0, 1, depth)
        run(5120, 32, 1005, 3, 5120 - 256, 1, 32)
        run(5120, 32, 1005, 3, 5120 - 256, 1, 40)
        run(5120, 64, 1005, 4, 0, 1, 16)
        run(5120, 64, 1005, 4, 0, 1, 24)
        run(5120, 64, 1005, 4, 0, 1, 32)
        run(5120, 16, 1005, 4, 0, 1, 40)
        run(5120, 16, 1005, 4, 0, 2, 24)
        run(5120, 8, 1005, 4, 0, 1, 40)
        run(5120, 8, 1005, 4, 0, 4, 24)
        run(5120, 8, 1005, 4, 0, 8, 16)
    if which == "small":
        for B in (1, 2, 4, 8):
            for nch, depth in [(1, 32), (4, 8), (8, 8), (16, 8), (32, 8), (64, 8)]:
                run(5120, B, 1005, 4, 0, nch, depth)
        for nch, depth in [(1, 32), (8, 8), (32, 8), (64, 8)]:
            run(2000, 1, 1005, 3, 1900, nch, depth)
    if which == "depth":      # round 2: shallow rings (HBM-bound launches do better with FEWER rows in flight)
        for depth in (3, 4, 6, 8, 16, 32):
            run(5120, 32, 1005, 3, 0, 1, depth)
        for depth in (3, 4, 6, 8, 16, 32):
            run(5120, 32, 1005, 4, 0, 1, depth)
        for depth in (3, 4, 6, 8, 32):
            run(5120, 32, 1005, 3, 5120 - 256, 1, depth)
        for depth in (3, 4, 6, 8, 16):
            run(4096, 128, 1000, 3, 0, 1, depth)
        for B in (16, 8):
            for depth in (3, 4, 6, 8, 16, 32):
                run(5120, B, 1005, 3, 0, 1, depth)
        for depth in (3, 4, 6, 8, 16, 32):
            run_state(4864, 32, 1005, depth)
    if which == "depth2":     # stores left out of the counted waits (DMN_VMCNT_COUNT_STORES = 0): re-tune
        for k in (3, 4):
            for depth in (8, 16, 24, 32):
                run(5120, 32, 1005, k, 0, 1, depth)
        for k in (3, 4):
            for depth in (4, 6, 8, 16, 24):
                run(2000, 64, 1005, k, 0, 1, depth, want_c=True)
        for depth in (4, 6, 8, 16):
            run(5120, 64, 1005, 3, 0, 1, depth)
        for depth in (16, 24, 32):
            run(5120, 16, 1005, 3, 0, 1, depth)
        for depth in (3, 4, 6):
            run(4096, 128, 1000, 3, 0, 1, depth)
    if which == "policy":
        run(5120, 32, 1005, 4, 0, 1, 32)
        run(5120, 32, 1005, 3, 0, 1, 32)
        run(5120, 32, 1005, 3, 5120 - 256, 1, 32)
        run(4096, 128, 1000, 3, 0, 1, 8)
        run(4096, 128, 1000, 3, 0, 1, 16)
    if which in ("all", "b1"):
        for nch, depth in [(1, 24), (8, 8), (16, 8), (32, 8), (64, 8), (128, 8)]:
            run(5120, 1, 1005, 4, 0, nch, depth)
