#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""bench.py -- the BASELINE.json metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--workload dmn512|msru_stress]

Workload `dmn512` (default) = BASELINE.json configs[1]: DMN high-res 512 px inference
(dpn92, num_filters 10, lang_layers 3, mix_we, upsampling amplification 32, random init,
20-token synthetic queries), batch 32 per GPU, inputs resident in HBM before the timed
region.  A step = one forward over one batch.  N > 1: one process per GPU, the batch of
(image, query) pairs is sharded, no data-path collective ("weak" scaling: 32 per GPU, so
N = 8 is configs[2]'s 256).  value = total queries / max-over-ranks wall time.  The ranks come
from torchrun's environment; started WITHOUT it (`python bench.py --gpus N`), this process
starts the N ranks itself as a `torch.distributed.run` child before it makes any GPU call, and
fails loudly when the box has fewer than N devices.  `n_gpus` in the line is
dist.get_world_size() and `collective_ranks` the result of a one-element all-reduce over the
job's backend (RCCL), i.e. the number of ranks the collective library really saw.

The same JSON line carries
  roofline     : the mSRU scan named by the metric -- the longest HBM-bound scan launch of the
                 step (layer 1, k = 3, T = 5120; layer 0's input is built inside its scan at this
                 batch size and no longer streamed): algorithmic bytes per launch / average
                 launch duration measured live with HIP events on the launch stream during the
                 timed steps, vs the 8 TB/s HBM peak; `hip_kernels` lists every entry point the
                 same way (plus TFLOP/s for the ones that are issue-bound, not HBM-bound);
  gemm_table_entries : shapes served from the recorded rocBLAS/hipBLASLt solution table
                 (dms_amd/tuning/, look-up only; --no-gemm-table to measure without it);
  shards       : configs[2]'s per-GPU shards of the global batch 256 -- B = 64 (4 GPUs) and B = 128
                 (2 GPUs) -- run on this ONE GPU after the timed region: queries/s each and the
                 difference of samples shared with the B = 32 batch (N = 1 only);
  train320     : configs[3] on this GPU, bounded (2 warm-up + 3 timed steps at B = 64): ms/step,
                 samples/s, the backward scan's roofline fraction, loss (N = 1 only; the whole
                 record comes from --workload train320);
  cpu_baseline : the CPU oracle (oracle/: plain-C scan + pure-torch reference-shaped glue +
                 torch-CPU DPN92) on a bounded sample of the same workload (a few batch-1
                 queries) on this box's host cores -- rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
SEED = 1111                    # reference default seed (train.py:75)


# the mSRU entry points: timed with HIP events inside the timed region (the roofline kernel is one of them)
MSRU_ENTRY_POINTS = frozenset(["dmn_sru_fwd", "dmn_sru_fwd_scan", "dmn_sru_state_fwd", "dmn_mix_sru_fwd",
                               "dmn_sru_fwd_image"])


class LaunchTimer(object):
    """Brackets every libdmn_hip entry-point call with HIP events on the launch stream."""

    def __init__(self):
        self.records = []
        self.enabled = False
        self.only = None                 # None: every entry point; else a set of entry-point names

    def begin(self, name, args):
        if not self.enabled or (self.only is not None and name not in self.only):
            return None
        a = torch.cuda.Event(enable_timing=True)
        a.record()
        return (name, args, a)

    def end(self, token):
        if token is None:
            return
        name, args, a = token
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        self.records.append((name, args, a, b))

    @staticmethod
    def describe(name, args):
        """(variant label, algorithmic bytes of this launch) -- SURVEY.md section 8(d)."""
        if name in ("dmn_sru_fwd", "dmn_sru_fwd_scan"):
            T, B, d, k = args[7:11]
            h_t0 = args[13]
            rd = (4 * k + (4 if k == 3 else 0)) * T * B * d
            wr = 4 * (T - h_t0) * B * d
            if name == "dmn_sru_fwd" and args[5]:           # c_all stored (training)
                wr += 4 * T * B * d
            label = "sru_fwd k=%d T=%d cols=%d%s%s" % (
                k, T, B * d, " tail" if h_t0 else "", " scan" if name.endswith("scan") else "")
            return label, rd + wr
        if name == "dmn_sru_fwd_image":
            T, B, d, k = args[7:11]
            return ("sru_fwd k=%d T=%d cols=%d -> (B, d, H, W) image" % (k, T, B * d),
                    (4 * k + (4 if k == 3 else 0) + 4) * T * B * d)
        if name == "dmn_sru_bwd":
            T, B, d, k = args[12:16]
            per = (4 * k + (4 if k == 3 else 0) + 8) + (4 * k + (4 if k == 3 else 0))
            return "sru_bwd k=%d T=%d cols=%d" % (k, T, B * d), per * T * B * d
        if name == "dmn_dynfilter_bwd":
            B, L, K, C, S = args[5:10]
            return "dynfilter_bwd", 4 * B * (2 * C * S + 2 * L * K * C + 2 * L * K * S)
        if name == "dmn_mix_bcast_add_bwd":
            # algorithmic: gQ read ONCE, gA / gBt / gP written, P read, gWp written once per sample
            B, L, K, S, M = args[9:14]
            return "mix_bcast_add_bwd", 4 * B * (S * L * M + S * M + L * M + 2 * L * K * S + M * K)
        if name == "dmn_cat_upsample2x_bwd":
            B, Cx, Cf, H, W = args[3:8]
            return "cat_upsample2x_bwd %dx%d" % (H, W), 4 * B * (Cx + Cf) * H * W * 5
        if name in ("dmn_dynfilter_fwd", "dmn_dynfilter_fwd_ws"):
            B, L, K, C, S = args[3:8] if name == "dmn_dynfilter_fwd" else args[4:9]
            return "dynfilter_fwd", 4 * B * (C * S + L * K * C + L * K * S)
        if name == "dmn_mix_sru_fwd":
            B, L, K, S, d = args[9:14]
            return ("mix+sru_fwd fused k=4 T=%d cols=%d" % (S * L, B * d),
                    4 * B * (S * 4 * d + L * 4 * d + L * K * S + S * L * d))
        if name == "dmn_mix_bcast_add":
            B, L, K, S, M = args[6:11]
            return "mix_bcast_add", 4 * B * (S * M + L * M + L * K * S + S * L * M)
        if name == "dmn_bias_sigmoid":
            rows, cols = args[2:4]
            return "bias_sigmoid (filter bank)", 8 * rows * cols
        if name == "dmn_cat_coords":
            B, C, H, W = args[2:6]
            return "cat_coords", 4 * B * H * W * (2 * C + 8)
        if name == "dmn_cat_upsample2x":
            B, Cx, Cf, H, W = args[3:8]
            return "cat_upsample2x %dx%d" % (H, W), 4 * B * (Cx + Cf) * H * W * 5
        if name == "dmn_wino_cat_upsample2x_transform":
            B, Cx, Cf, H, W, m = args[3:9]
            tiles = B * (2 * H // m) * (2 * W // m)
            return ("wino_in F(%dx%d) cat+up+B^T d B %dx%d" % (m, m, H, W),
                    4 * (Cx + Cf) * (B * H * W + (m + 2) ** 2 * tiles))
        if name == "dmn_wino_output_transform":
            B, Co, H, W, m = args[3:8]
            tiles = B * (2 * H // m) * (2 * W // m)
            return ("wino_out F(%dx%d) A^T M A %dx%d" % (m, m, H, W),
                    4 * Co * ((m + 2) ** 2 * tiles + B * 4 * H * W))
        if name == "dmn_sru_state_fwd":
            T, B, d = args[4:7]
            return "sru_state (u0|f only) T=%d cols=%d" % (T, B * d), 8 * T * B * d
        if name == "dmn_sru_state_fwd_train":
            T, B, d = args[5:8]
            return "sru_state fwd +c (u0|f only) T=%d cols=%d" % (T, B * d), 12 * T * B * d
        if name == "dmn_sru_state_bwd":
            T, B, d = args[9:12]
            return "sru_state bwd (u0|f only) T=%d cols=%d" % (T, B * d), 20 * T * B * d
        if name == "dmn_gconv3x3_bn_relu":
            N, C, H, W = args[9:13]
            st = args[14]
            return "dpn gconv3x3+2x(bn,relu) (all shapes)", 4 * N * C * H * W * (st * st + 1) // (st * st)
        if name == "dmn_bn_relu_s2":
            N, C, H, W = args[5:9]
            return "dpn bn_relu stride 2", 4 * N * C * H * W * 3 // 4
        if name == "dmn_head_final":
            B, Cx, Cf, H, W = args[5:10]
            return "head_final (cat+up+conv C->1) %dx%d" % (H, W), 4 * B * H * W * (Cx + Cf + 4)
        if name == "dmn_tap_upsample_gather":
            B, Co, H, W = args[3:7]
            return "tap_gather (9 planes -> up x2) %dx%d" % (H, W), 4 * B * Co * H * W * (9 + 4)
        if name == "dmn_tap_project":
            B, Cx, Cf, HW = args[4:8]
            return "tap_project (C -> 9 rows)", 4 * B * HW * (Cx + Cf + 9)
        if name == "dmn_channel_dot":
            N, C, HW = args[4:7]
            return "channel_dot C=%d" % C, 4 * N * HW * (C + 1)
        if name == "dmn_bn_relu_maxpool":
            N, C, H, W = args[5:9]
            return "stem bn_relu+maxpool", 4 * N * C * H * W * 5 // 4
        if name == "dmn_bn_relu":
            N, C, HW = args[5:8]
            return "dpn bn_relu (all shapes)", 8 * N * C * HW
        if name == "dmn_dpn_stage_pass":
            N, rows, c1c, inc, Co, HW = args[6:12]
            return ("dpn stage pass (in-place merge + bn_relu, all shapes)",
                    4 * N * HW * (2 * Co + 2 * inc + (Co if args[2] else 0)))
        if name == "dmn_dpn_merge":
            N, Cb, Cy, c1c, HW = args[7:12]
            Co = Cb + Cy - c1c
            return "dpn merge+bn_relu (all shapes)", 4 * N * HW * (Cb + Cy + Co * (2 if args[3] else 1))
        return name, 0

    @staticmethod
    def flops(name, args):
        """fp32 flops of a launch for the kernels whose arithmetic intensity puts them above the
        HBM ridge (reported next to the HBM fraction so that it is not misread), else 0."""
        if name == "dmn_gconv3x3_bn_relu":
            N, C, H, W, groups, st = args[9:15]
            return 2 * N * C * (H // st) * (W // st) * (C // groups) * 9
        if name in ("dmn_dynfilter_fwd", "dmn_dynfilter_fwd_ws"):
            B, L, K, C, S = args[3:8] if name == "dmn_dynfilter_fwd" else args[4:9]
            return 2 * B * L * K * C * S
        if name == "dmn_mix_sru_fwd":
            B, L, K, S, d = args[9:14]
            return 2 * B * S * L * d * 4 * K
        if name == "dmn_dynfilter_bwd":                      # two thin GEMMs: gF = gP . V^T and gV = F^T . gP
            B, L, K, C, S = args[5:10]
            return 2 * 2 * B * L * K * C * S
        return 0

    @staticmethod
    def issue_bound_ms(name, args):
        """Kernels bound by one wave's in-order instruction issue, not by HBM or a FLOP peak: the time their
        instruction stream needs at the documented issue rates, or 0.
        dmn_mix_sru_fwd (matrix-core pair kernel, K = 10, L = 20): per block of 32 steps the OUTPUT wave issues
        20 fp32 MFMAs (64 cycles each in the pipe), 32 v_permlane32_swap (8), 16 ds_read_b128 + 32 v_pk_add_f32 for
        the tile start values (8 / 4), the scan's 3 transcendentals (quarter rate: 16) + 6 full-rate VALU (4) per
        step, ~150 cycles of operand hand-over: 4246 cycles.  Every 64-column window is one wave pair, all pairs run
        at once while there are at most 1024 waves (one per SIMD), in ceil(waves / 1024) shares of a SIMD beyond."""
        if name == "dmn_mix_sru_fwd":
            B, L, K, S, d = args[9:14]
            if K != 10 or L != 20:
                return 0.0
            blocks = (S * L + 31) // 32
            cycles = 20 * 64 + 32 * 8 + (16 * 8 + 32 * 4) + 32 * (3 * 16 + 6 * 4) + 150
            waves = 2 * B * ((d + 15 + 63) // 64)
            share = max((waves + 1023) // 1024, 1)
            return share * blocks * cycles / 2.4e9 * 1e3
        if name == "dmn_mix_bcast_add_bwd":
            # one-pass kernel (K <= 10, L <= 20, 512-column workgroups, one per CU = two waves per SIMD): 437 VALU
            # instructions per wave and pixel (rocprofv3 SQ_INSTS_VALU, profiles/r3_mix_bwd_one.txt) at ~4.4 cycles
            # (packed fp32 FMAs, DPP / permlane exchanges); its HBM floor is lower (2.2 GB: 0.28 ms)
            B, L, K, S, M = args[9:14]
            wgs = B * ((M + 511) // 512)
            if K > 10 or L > 20 or M % 4 or wgs < 256:
                return 0.0
            return ((wgs + 255) // 256) * S * 2 * 437 * 4.4 / 2.4e9 * 1e3
        return 0.0

    def summary(self):
        agg = {}
        for name, args, a, b in self.records:
            label, nbytes = self.describe(name, args)
            e = agg.setdefault(label, {"calls": 0, "ms": 0.0, "bytes": 0, "flops": 0, "issue_ms": 0.0})
            e["calls"] += 1
            e["ms"] += a.elapsed_time(b)
            e["bytes"] += nbytes
            e["flops"] += self.flops(name, args)
            e["issue_ms"] += self.issue_bound_ms(name, args)
        out = []
        for label, e in agg.items():
            avg_ms = e["ms"] / e["calls"]
            gbs = (e["bytes"] / e["calls"]) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            out.append({"kernel": label, "calls": e["calls"], "avg_ms": round(avg_ms, 4),
                        "total_ms": round(e["ms"], 3),
                        "algorithmic_bytes_per_launch": e["bytes"] // e["calls"],
                        "achieved_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4)})
            if e["issue_ms"]:
                out[-1].update({"bound": "wave issue (one in-order instruction stream per SIMD; neither HBM nor a FLOP peak)",
                                "issue_bound_ms": round(e["issue_ms"] / e["calls"], 4),
                                "frac_of_bound": round(e["issue_ms"] / e["ms"], 4)})
            elif e["flops"]:
                tf = e["flops"] / (e["ms"] * 1e-3) / 1e12
                out[-1].update({"achieved_TFLOPs": round(tf, 1), "frac_of_157TF_fp32": round(tf / 157.0, 4),
                                "bound": "fp32 issue (above the HBM ridge)"})
        out.sort(key=lambda r: -r["total_ms"])
        return out


def measured_traffic(kernel_label):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic_r3.json, else r2 / r1; separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH x2 correction) or None."""
    for name in ("traffic_r3.json", "traffic_r2.json", "traffic_r1.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                table = json.load(f)
            return table[kernel_label]["hbm_bytes"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def build_model(device, high_res=True):
    import dms_amd
    torch.manual_seed(SEED)
    net = dms_amd.DMN(1000, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                      high_res=high_res, upsampling_mode="bilinear",
                      upsampling_amplification=32)
    return net.to(device).eval()


def serving_latency(net, device, reps=30):
    """Batch-1 latency of the same model (the reference's own serving shape, dmn.py:142):
    eager launches vs one captured HIP graph (dms_amd.GraphedDMN).  Extra information, not
    part of `value`."""
    import dms_amd
    g = torch.Generator(device=device).manual_seed(SEED)
    img = torch.randn((1, 3, 512, 512), device=device, generator=g)
    words = torch.randint(0, 1000, (1, 20), device=device, generator=g)

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    with torch.no_grad():
        eager = timeit(lambda: net(img, words))
        graphed = dms_amd.GraphedDMN(net, 1, 512, 512, 20)
        replay = timeit(lambda: graphed(img, words))
        same = bool(torch.equal(graphed(img, words), net(img, words)))
    return {"workload": "batch-1 512px high-res query", "eager_ms": round(eager, 3),
            "hip_graph_ms": round(replay, 3), "graph_equals_eager": same}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_time(fn, reps):
    times = []
    for i in range(reps + 1):
        t0 = time.perf_counter()
        out = fn()
        if i:                                              # first pass is warm-up
            times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], out


def cpu_baseline_dmn(net, reps, threads):
    """CPU oracle (= the reference path's arithmetic: torch-CPU DPN92 + oracle.dmn_ref's reference-shaped
    hot path with the plain-C SRU scan + reference-shaped upsampling) on this box's host cores, on
    bounded samples of SURVEY 8(d)'s cases: 512 px high-res at B = 1 (the headline `value`) and B = 8,
    configs[0] (320 px low-res, B = 1), and the SRU-only scan of configs[4] scaled to B = 8."""
    import numpy as np
    from oracle import dmn_ref, sru_ref
    torch.set_num_threads(threads)
    sru_ref.set_threads(threads)
    cpu_net = build_model("cpu")
    torch.nn.Module.load_state_dict(cpu_net, {k: v.cpu() for k, v in net.state_dict().items()})
    P = dict(cpu_net.state_dict())
    cfg = dict(num_filters=10, vis_size=2688, lang_layers=3, mixed_layers=3, mix_we=True,
               high_res=True)
    g = torch.Generator().manual_seed(SEED)
    img = torch.randn(1, 3, 512, 512, generator=g)
    words = torch.randint(0, 1000, (1, 20), generator=g)

    def query(net_, P_, cfg_, im, wd):
        vis, feats = net_.langvis.base(im)
        x = dmn_ref.hot_path(P_, vis, wd, cfg_, use_c=True)
        return dmn_ref.upsampling(P_, x, feats, 5) if cfg_["high_res"] else x

    with torch.no_grad():
        med, out = _median_time(lambda: query(cpu_net, P, cfg, img, words), reps)
        # 512 px high-res, batch 8 (backbone and head batched, hot path per sample as in the reference)
        img8 = torch.randn(8, 3, 512, 512, generator=g)
        words8 = torch.randint(0, 1000, (8, 20), generator=g)
        med8, _ = _median_time(lambda: query(cpu_net, P, cfg, img8, words8), 2)
        # configs[0]: 320 px low-res, batch 1 (the reference's own CPU-runnable case)
        low = build_model("cpu", high_res=False)
        Pl = dict(low.state_dict())
        cfgl = dict(cfg, high_res=False)
        img3 = torch.randn(1, 3, 320, 320, generator=g)
        medl, _ = _median_time(lambda: query(low, Pl, cfgl, img3, words), 3)
    # SRU-only scan, configs[4] scaled to B = 8: T = 4096, d = 1000, k = 3
    T, B, d = 4096, 8, 1000
    rng = np.random.default_rng(SEED)
    u = rng.standard_normal((T, B, d, 3), dtype=np.float32)
    xx = rng.standard_normal((T, B, d), dtype=np.float32)
    bias = (0.1 * rng.standard_normal(2 * d)).astype(np.float32)
    meds, _ = _median_time(lambda: sru_ref.sru_fwd_c(u, xx, bias, None, "tanh", 1.0, want_c=False), 3)
    scan_gbs = 20.0 * T * B * d / meds / 1e9
    cb = {"value": round(1.0 / med, 4), "unit": "queries/s", "cores": threads, "kind": "port",
          "host_cpus": os.cpu_count(), "cpu_model": _cpu_model(), "threads_used": threads,
          "sample": "%d batch-1 512px high-res queries (median of %d after 1 warm-up): torch-CPU "
                    "DPN92 + oracle/dmn_ref hot path (C scan, OpenMP) + oracle upsampling"
                    % (reps, reps),
          "seconds_per_query": round(med, 3),
          "cases": {
              "dmn512_highres_b1": {"queries_per_s": round(1.0 / med, 4), "reps": reps},
              "dmn512_highres_b8": {"queries_per_s": round(8.0 / med8, 4), "reps": 2},
              "dmn320_lowres_b1 (configs[0])": {"queries_per_s": round(1.0 / medl, 4), "reps": 3},
              "sru_scan_T4096_B8_d1000_k3": {"GBs_algorithmic_20B": round(scan_gbs, 2),
                                             "ms": round(meds * 1e3, 2), "reps": 3}}}
    return cb, (img, words, out)


def run_train320(args, device, rank, world, timer, quiet=False):
    """BASELINE configs[3]: low-res 320 px training step, batch 64 per GPU: forward -> bilinear
    upsample of the logits to the mask size -> BCEWithLogits vs random {0,1} masks
    (train.py:275-283) -> backward -> gradient all-reduce (RCCL) -> Adam(lr 1e-5) (train.py:71,242)."""
    import threading
    import torch.nn.functional as F
    from dms_amd.dist import OverlappedGradReducer

    # heartbeat: the first step compiles / finds MIOpen kernels for every DPN92 convolution (forward,
    # backward-data, backward-weights); a silent process is taken for hung.  (MIOpen's find mode is left
    # at its default: MIOPEN_FIND_MODE=FAST costs 60 ms per step in worse convolution picks.)
    phase = {"name": "building the model", "t0": time.perf_counter(), "stop": False}

    def heartbeat():
        while not phase["stop"]:
            time.sleep(20)
            if not phase["stop"] and not quiet:
                print("[train320] rank %d alive: %s (%.0f s)" % (rank, phase["name"], time.perf_counter() - phase["t0"]),
                      file=sys.stderr, flush=True)

    threading.Thread(target=heartbeat, daemon=True).start()
    # watchdog: a step that exceeds its limit dumps every thread's Python stack to stderr and exits
    # non-zero, so a hung exchange names itself instead of running into the launcher's time limit.
    # DMN_BENCH_WATCHDOG="first,later" in seconds (the first step includes the MIOpen find phase)
    import faulthandler
    wd_first, wd_later = [float(v) for v in os.environ.get("DMN_BENCH_WATCHDOG", "900,240").split(",")]

    def arm(limit):
        faulthandler.cancel_dump_traceback_later()
        if limit > 0:
            faulthandler.dump_traceback_later(limit, exit=True)
    net = build_model(device, high_res=False).train()
    # same Adam (train.py:242) through torch's single-kernel implementation: 2.6 -> <1 ms per step
    opt = torch.optim.Adam(net.parameters(), lr=1e-5, fused=True)
    # gradients live in flat buckets; each bucket's RCCL all-reduce starts from an autograd hook while
    # backward is still running (dms_amd/dist.py)
    red = OverlappedGradReducer(list(net.parameters()), bucket_mb=128, average=True) if world > 1 else None
    B = args.batch
    g = torch.Generator(device=device).manual_seed(SEED + rank)
    img = torch.randn((B, 3, 320, 320), device=device, generator=g)
    words = torch.randint(0, 1000, (B, 20), device=device, generator=g)
    mask = (torch.rand((B, 1, 320, 320), device=device, generator=g) > 0.5).float()
    crit = torch.nn.BCEWithLogitsLoss()
    ar_ms = []

    def step():
        if red is not None:
            red.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        out = net(img, words)
        out = F.interpolate(out, size=mask.shape[-2:], mode="bilinear", align_corners=False)
        loss = crit(out, mask)
        loss.backward()
        if red is not None:
            # what is left of the exchange after backward has returned (the exposed part)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            red.finish()
            b.record()
            ar_ms.append((a, b))
        opt.step()
        return loss

    for i in range(args.warmup):
        phase["name"] = "warm-up step %d (MIOpen find / kernel compilation on the first one)" % i
        print("[train320] rank %d starting warm-up step %d" % (rank, i), file=sys.stderr, flush=True)
        arm(wd_first if i == 0 else wd_later)
        tw = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        print("[train320] rank %d warm-up step %d done (%.1f s)" % (rank, i, time.perf_counter() - tw),
              file=sys.stderr, flush=True)
    phase["name"] = "timed steps"
    arm((wd_later * args.steps) if args.warmup else (wd_first + wd_later * args.steps))
    if world > 1:
        dist.barrier()
    timer.records, ar_ms[:] = [], []
    if red is not None:
        red.launch_log = []
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    phase["stop"] = True
    arm(0)
    assert bool(torch.isfinite(loss))
    el = torch.tensor([elapsed], dtype=torch.float64,
                      device=device if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    kernels = timer.summary()
    bw = [r for r in kernels if r["kernel"].startswith("sru_bwd") and "T=2000" in r["kernel"]]
    dom = bw[0] if bw else kernels[0]
    nparams = sum(p.numel() for p in net.parameters())
    line = {"metric": "training samples/sec (whole node), DMN low-res 320px step; mSRU bwd achieved HBM GB/s",
            "value": round(world * B * args.steps / elapsed, 3), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DMN low-res 320px training step (BASELINE configs[3]): BCEWithLogits, "
                                   "Adam lr 1e-5, train() mode", "batch_per_gpu": B,
                       "global_batch": B * world, "grad_allreduce_MB": round(nparams * 4 / 1e6, 1),
                       "parallelism": "data parallel; gradients in flat 128 MB buckets, each bucket's all-reduce "
                                      "launched from an autograd hook during backward"},
            "roofline": {"bound": "hbm", "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": dom["frac_of_8TBs"], "traffic": measured_traffic(dom["kernel"]),
                         "kernel": dom["kernel"], "avg_launch_ms": dom["avg_ms"],
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"]},
            "allreduce_exposed_ms": round(sum(a.elapsed_time(b) for a, b in ar_ms) / max(len(ar_ms), 1), 3) if ar_ms else None,
            "allreduce_buckets_launched_during_backward": (round(sum(1 for _, inb in red.launch_log if inb) / float(args.steps), 2) if red is not None else None),
            "allreduce_buckets": (len(red.buckets) if red is not None else None),
            "hip_kernels_ms_per_step": round(sum(r["total_ms"] for r in kernels) / args.steps, 3),
            "hip_kernels": kernels, "loss": float(loss)}
    line.update(_world_proof(device, world))
    if red is not None:
        red.remove()
    return line


def launch_ranks(n):
    """`python bench.py --gpus N` without torchrun's environment: start the N ranks from here, as a
    `python -m torch.distributed.run` child (the driver's own form of the command), BEFORE this process
    has made any GPU call -- torch.cuda.device_count() only counts -- and leave with the child's exit
    code.  Fewer than N devices on the box is an error (unless the one-GPU rehearsal hooks
    DMN_BENCH_DEVICE / DMN_BENCH_BACKEND pin every rank to one card on purpose)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if "DMN_BENCH_DEVICE" not in os.environ and have < n:
        raise SystemExit("bench.py --gpus %d: this box has %d GPU(s); refusing to report an %d-GPU number "
                         "from fewer devices" % (n, have, n))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def _world_proof(device, world):
    """What the JSON line says about the job's size comes from the process group itself: n_gpus =
    dist.get_world_size(), collective_ranks = a one-element sum all-reduce over the job's backend."""
    if not (dist.is_available() and dist.is_initialized()):
        if world != 1:
            raise SystemExit("bench.py: %d ranks expected but no process group exists" % world)
        return {"n_gpus": 1, "collective_ranks": 1, "collective_backend": None}
    backend = dist.get_backend()
    one = torch.ones(1, dtype=torch.float32, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    seen = int(round(float(one.item())))
    if seen != dist.get_world_size() or seen != world:
        raise SystemExit("bench.py: the all-reduce saw %d ranks, world size %d, --gpus %d"
                         % (seen, dist.get_world_size(), world))
    return {"n_gpus": dist.get_world_size(), "collective_ranks": seen,
            "collective_backend": "rccl" if backend == "nccl" else backend}


def run_shards(net, device, head4, steps=3, warmup=2):
    """BASELINE configs[2]'s per-GPU shards of the global batch 256 on THIS GPU: B = 64 (the 4-GPU
    shard) and B = 128 (the 2-GPU shard); B = 32 (8 GPUs) is the headline.  The first 4 samples of
    each batch are the headline batch's own first 4 (same image, same query), so their logits must
    agree with the B = 32 run to GEMM-kernel-choice rounding (the hand-written kernels are bitwise
    batch-invariant, tests/test_gpu_parity_full.py); oracle parity inside such batches is
    tests/test_gpu_shards.py."""
    res = {}
    img4, words4, out4 = head4
    for B in (64, 128):
        g = torch.Generator(device=device).manual_seed(SEED + B)
        img = torch.randn((B, 3, 512, 512), device=device, generator=g)
        words = torch.randint(0, 1000, (B, 20), device=device, generator=g)
        img[:4], words[:4] = img4, words4
        with torch.no_grad():
            for _ in range(warmup + 1):
                out = net(img, words)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = net(img, words)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        ok = tuple(out.shape) == (B, 1, 512, 512) and bool(torch.isfinite(out).all())
        rel = float((out[:4] - out4).abs().max() / out4.abs().max())
        res["B=%d" % B] = {"gpus_in_configs2": 256 // B, "queries_per_s": round(B / dt, 2),
                           "ms_per_step": round(dt * 1e3, 3), "steps": steps, "finite": ok,
                           "rel_diff_vs_same_samples_in_B32": rel,
                           "peak_mem_GB": round(torch.cuda.max_memory_allocated(device) / 1e9, 2)}
        del img, words, out
        torch.cuda.empty_cache()
    return res


def train320_child(steps=3, warmup=2, limit_s=420):
    """configs[3] as `bench.py --workload train320 --steps 3 --warmup 2` in a child process (started,
    not exec'ed) under a time limit; returns a summary of its JSON line."""
    import subprocess
    env = dict(os.environ, DMN_BENCH_WATCHDOG=os.environ.get("DMN_BENCH_WATCHDOG", "300,60"))
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "train320", "--steps", str(steps),
           "--warmup", str(warmup)]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return {"error": "no result within %d s" % limit_s}
    wall = time.perf_counter() - t0
    rows = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not rows:
        return {"error": "rc %d" % r.returncode, "stderr_tail": r.stderr.decode(errors="replace")[-600:]}
    t = json.loads(rows[-1])
    keep = ("value", "unit", "steps", "warmup", "ms_per_step", "loss", "hip_kernels_ms_per_step")
    out = {"workload": t["config"]["workload"], "batch_per_gpu": t["config"]["batch_per_gpu"]}
    out.update({k: t[k] for k in keep})
    out["sru_bwd"] = {k: t["roofline"][k] for k in ("kernel", "avg_launch_ms", "achieved", "frac")}
    out["hip_kernels"] = [{k: r_[k] for k in r_ if k in ("kernel", "calls", "avg_ms", "frac_of_8TBs",
                                                          "achieved_TFLOPs", "bound", "frac_of_bound")}
                          for r_ in t["hip_kernels"][:12]]
    out["child_wall_s"] = round(wall, 1)
    return out


def run_msru_stress(device, steps, warmup, timer):
    """BASELINE.json configs[4]: standalone scan, T = 64*64, hidden 1000, batch 128, k = 3."""
    from dms_amd import ops
    T, B, d, k = 4096, 128, 1000, 3
    g = torch.Generator(device=device).manual_seed(SEED)
    u = torch.randn((T, B, d, k), device=device, generator=g)
    x = torch.randn((T, B, d), device=device, generator=g)
    bias = 0.1 * torch.randn((2 * d,), device=device, generator=g)
    for _ in range(warmup):
        ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, 0, want_c=False, nchunks=1)
    torch.cuda.synchronize()
    timer.records = []
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(steps):
        ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, 0, want_c=False, nchunks=1)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    timer.enabled = False
    rows = timer.summary()
    return rows[0], wall / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="(image, query) pairs per GPU (default 32; 64 for train320)")
    ap.add_argument("--workload", default="dmn512", choices=["dmn512", "msru_stress", "train320"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=7)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the objects measured after the timed region (shards, train320, serving_b1, msru_stress)")
    ap.add_argument("--no-head-link", action="store_true",
                    help="A/B switch: compact pyramid taps and the two-GEMM head levels instead of the "
                         "backbone writing each tap into the head's [x | feat] buffer")
    ap.add_argument("--no-gemm-table", action="store_true",
                    help="do not load the recorded per-shape BLAS solution table (dms_amd/tuning/)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 64 if args.workload == "train320" else 32

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)                      # never returns; no GPU call has been made yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    # rehearsal hooks (one-GPU box): DMN_BENCH_DEVICE pins every rank to one card and
    # DMN_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on the same device
    dev_index = int(os.environ.get("DMN_BENCH_DEVICE", local_rank))
    backend = os.environ.get("DMN_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from dms_amd import _lib
    _lib.load()
    timer = LaunchTimer()
    _lib.TIMER = timer

    def barrier():
        if world > 1:
            dist.barrier()

    if args.workload == "train320":
        line = run_train320(args, device, rank, world, timer)
        if rank == 0:
            print(json.dumps(line))
        if world > 1:
            dist.destroy_process_group()
        return

    if args.workload == "msru_stress":
        row, ms = run_msru_stress(device, args.steps, args.warmup, timer)
        line = {"metric": "mSRU achieved HBM GB/s (standalone scan, algorithmic 20 B per element)",
                "value": row["achieved_GBs"], "unit": "GB/s", "n_gpus": 1, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "msru_stress T=4096 B=128 d=1000 k=3 (BASELINE configs[4])"},
                "roofline": {"bound": "hbm", "achieved": row["achieved_GBs"], "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": row["frac_of_8TBs"],
                             "traffic": measured_traffic(row["kernel"]),
                             "kernel": row["kernel"], "avg_launch_ms": row["avg_ms"]}}
        if rank == 0:
            print(json.dumps(line))
        return

    from dms_amd import ops as dmn_ops
    gemm_table_entries = 0 if args.no_gemm_table else dmn_ops.use_gemm_table()
    net = build_model(device)
    if args.no_head_link:
        net.langvis.base.tap_prefix = None
    B = args.batch
    g = torch.Generator(device=device).manual_seed(SEED + rank)
    img = torch.randn((B, 3, 512, 512), device=device, generator=g)
    words = torch.randint(0, 1000, (B, 20), device=device, generator=g)

    with torch.no_grad():
        # one priming pass outside warm-up and timing: lazy initialisation only (MIOpen's find for
        # the stem conv, the per-shape BLAS pick, cached weight layouts), so that the result does
        # not depend on the caller's --warmup
        out = net(img, words)
        for _ in range(args.warmup):
            out = net(img, words)
        torch.cuda.synchronize()
        barrier()
        # inside the timed region only the mSRU launches (the `roofline` kernel among them) are bracketed
        # with HIP events; the table of all ~150 entry-point calls per step comes from extra steps after it
        # (300 event records per step cost the step ~0.5 %: DMN_BENCH_TIMER_ALL=1 brings them back)
        timer.records = []
        timer.only = None if os.environ.get("DMN_BENCH_TIMER_ALL") == "1" else MSRU_ENTRY_POINTS
        timer.enabled = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = net(img, words)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        timer.enabled = False
        timed_records = timer.records
        table_steps = args.steps
        if timer.only is not None:
            timer.records, timer.only, timer.enabled = [], None, True
            table_steps = min(args.steps, 5)
            for _ in range(table_steps):
                out = net(img, words)
            torch.cuda.synchronize()
            timer.enabled = False
    assert tuple(out.shape) == (B, 1, 512, 512) and bool(torch.isfinite(out).all())
    head4 = (img[:4].clone(), words[:4].clone(), out[:4].clone())
    proof = _world_proof(device, world)

    el = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    qps = world * B * args.steps / elapsed

    kernels = timer.summary()                       # every entry point (the extra steps, or the timed region)
    timer.records = timed_records
    timed = timer.summary()                         # the timed region's own records: roofline comes from here
    scans = [r for r in timed if r["kernel"].startswith("sru_fwd") and "T=%d" % (256 * 20) in r["kernel"]]
    dom = scans[0] if scans else timed[0]
    roofline = {"bound": "hbm", "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": dom["frac_of_8TBs"], "traffic": measured_traffic(dom["kernel"]),
                "kernel": dom["kernel"], "avg_launch_ms": dom["avg_ms"],
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"]}
    hip_ms = sum(r["total_ms"] for r in kernels) / table_steps

    line = {"metric": "queries/sec (whole node) high-res 512px DMN inference; mSRU achieved HBM GB/s",
            "value": round(qps, 3), "unit": "queries/s", "n_gpus": proof["n_gpus"], "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "DMN high-res 512px inference (BASELINE configs[1]): dpn92, "
                                   "num_filters 10, lang_layers 3, mix_we, amplification 32, "
                                   "20-token queries, random init, eval()",
                       "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "batch-sharded replicas, no collective"},
            "collective_ranks": proof["collective_ranks"], "collective_backend": proof["collective_backend"],
            "roofline": roofline,
            "msru_GBs": dom["achieved_GBs"],
            "gemm_table_entries": gemm_table_entries,
            "hip_kernels_ms_per_step": round(hip_ms, 3),
            "hip_kernels_from": ("the timed region" if os.environ.get("DMN_BENCH_TIMER_ALL") == "1" else
                                 "%d extra steps after the timed region (inside it only the mSRU launches carry "
                                 "HIP events)" % table_steps),
            "hip_kernels": kernels}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = min(os.cpu_count() or 1, 16)
        cb, (cimg, cwords, cout) = cpu_baseline_dmn(net, args.cpu_reps, threads)
        # parity of the path that was just timed: the oracle's query sits in a batch of the
        # benchmarked size (shape-dependent choices such as the Winograd head levels are made
        # per batch), the other entries are the benchmark's own inputs
        with torch.no_grad():
            pimg, pwords = img.clone(), words.clone()
            pimg[0], pwords[0] = cimg[0].to(device), cwords[0].to(device)
            gout = net(pimg, pwords)[:1].cpu()
        cb["gpu_vs_oracle_rel_err"] = float((gout - cout).abs().max() / cout.abs().max())
        cb["mask_agreement"] = float(((gout > 0) == (cout > 0)).float().mean())
        line["cpu_baseline"] = cb
    if rank == 0 and world == 1 and not args.no_extras:
        line["serving_b1"] = serving_latency(net, device)
        # configs[4] microbench beside it (same process, after the timed region)
        row, ms = run_msru_stress(device, 20, 3, timer)
        line["msru_stress"] = {"workload": "T=4096 B=128 d=1000 k=3", "achieved_GBs": row["achieved_GBs"],
                               "frac_of_8TBs": row["frac_of_8TBs"], "avg_launch_ms": row["avg_ms"]}
        # configs[2]'s 4-GPU and 2-GPU shards on this one GPU
        try:
            line["shards"] = run_shards(net, device, head4)
        except Exception as e:                                        # the headline must survive
            line["shards"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # configs[3], bounded, in a fresh process of its own (this one's memory released first): a
        # hang or crash there cannot take the headline with it
        del net, img, out
        torch.cuda.empty_cache()
        line["train320"] = train320_child()
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
