#!/usr/bin/env python
"""profiles/traffic_r3.json from round 3's PMC passes (tools/profile_round3.sh):
per-launch HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB; FETCH_SIZE reports half of a wide streaming read
on gfx950 -- MI355X_MICROARCH.md, HBM section), median over the launches of a kernel, next to the
algorithmic bytes.  usage: python tools/pmc_summary_r3.py gpurun_out/prof_r3"""
import collections, csv, json, os, statistics, sys

root = sys.argv[1]


def medians(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "dmn::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in agg.items()}


fetch = medians(os.path.join(root, "r2f", "r2f_counter_collection.csv"))
write = medians(os.path.join(root, "r2w", "r2w_counter_collection.csv"))
fetch.update(medians(os.path.join(root, "scf", "scf_counter_collection.csv")))
write.update(medians(os.path.join(root, "scw", "scw_counter_collection.csv")))
fetch.update(medians(os.path.join(root, "r3f", "r3f_counter_collection.csv")))
write.update(medians(os.path.join(root, "r3w", "r3w_counter_collection.csv")))

E = 5120 * 32160                      # elements of the in-situ mSRU scans (B = 32, 512 px)
EB = 2000 * 64320                     # elements of the training scans (configs[3])
LABELS = {
    ("dmn::sru_fwd_cols<3, 1, 16, false>", 32192): ("sru_fwd k=3 T=5120 cols=32160", 20 * E),
    ("dmn::sru_fwd_cols<4, 1, 16, false>", 32192): ("sru_fwd k=4 T=5120 cols=32160", 20 * E),
    ("dmn::sru_fwd_cols<3, 1, 32, false>", 32192): ("sru_fwd k=3 T=5120 cols=32160 tail", 16 * E + 4 * 256 * 32160),
    # configs[4] goes out as two column ranges of 1000 wavefronts (dmn_sru_fwd): HALF the entry point's bytes per launch
    ("dmn::sru_fwd_cols<3, 1, 8, false>", 64000): ("sru_fwd k=3 T=4096 cols=128000", 20 * 4096 * 128000, 2),
    ("dmn::sru_fwd_cols<2, 0, 32, false>", 32192): ("sru_state (u0|f only) T=4864 cols=32160", 8 * 4864 * 32160 + 4 * 32160),
    ("dmn::sru_fwd_cols<3, 1, 8, true>", 64320): ("sru_fwd k=3 T=2000 cols=64320", 24 * EB),
    ("dmn::sru_fwd_cols<4, 1, 8, true>", 64320): ("sru_fwd k=4 T=2000 cols=64320", 24 * EB),
    ("dmn::sru_bwd_cols<3, 1, 4>", 64320): ("sru_bwd k=3 T=2000 cols=64320", 40 * EB),
    ("dmn::sru_bwd_cols<4, 1, 4>", 64320): ("sru_bwd k=4 T=2000 cols=64320", 40 * EB),
    ("dmn::mix_sru_pair_kernel<1>", 65536): ("mix+sru_fwd fused k=4 T=5120 cols=32160 (issue-bound), VALU pair kernel", 807208960),
    ("dmn::mix_sru_mfma_kernel<1>", 65536): ("mix+sru_fwd fused k=4 T=5120 cols=32160 (issue-bound)", 807208960),
    # round 3
    ("dmn::mix_bwd_one_kernel<2>", 262144): ("mix_bcast_add_bwd one pass B=64 S=100 L=20 K=10 M=4020 (gQ once)",
                                         4 * 64 * (100 * 20 * 4020 + 100 * 4020 + 20 * 4020 + 2 * 20 * 10 * 100 + 4020 * 10)),
    ("dmn::dynfilter_fwd_dma<64, 4>", 131072): ("dynfilter_fwd MFMA B=32 C=2696 S=256 (MFMA-bound; LDS-DMA ring kernel, whole contraction per workgroup)",
                                                 4 * 32 * (2696 * 256 + 200 * 2696 + 200 * 256)),
    ("dmn::bias_sigmoid_kernel<4>", 4423680): ("bias_sigmoid 640 x 26960 in place", 8 * 640 * 26960),
    ("dmn::sru_fwd_cols_img<3, 1, 16>", 32192): ("sru_fwd_image k=3 T=256 cols=32160 (h stored as the (B,Hm,H,W) map)", 20 * 256 * 32160),
}
out = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round3.sh over "
               "tools/pmc_round3.py, tools/pmc_round2.py and tools/pmc_scan.py). "
               "Units KiB, median of the launches; hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports "
               "half of a wide streaming read; WRITE_SIZE exact).  The fused mix+scan kernel re-reads small L2-resident "
               "operands (Bt, P, Wp) that the x2 rule over-counts: its WRITE_SIZE alone is +2.4 % over the h bytes.  "
               "The one-pass mix backward writes its 8 per-column-block gP partials (41 MB) and reads them back in mix_bwd_sum_parts: "
               "that workspace traffic is in hbm_bytes and not in algorithmic_bytes."}
for key, spec in LABELS.items():
    label, alg = spec[0], spec[1]
    launches = spec[2] if len(spec) > 2 else 1          # launches per entry-point call
    if key in fetch and key in write:
        hb = int(round((2 * fetch[key] + write[key]) * 1024)) * launches
        out[label] = {"fetch_size_kib": round(fetch[key], 1), "write_size_kib": round(write[key], 1),
                      "hbm_bytes": hb, "algorithmic_bytes": alg, "ratio": round(hb / alg, 4)}
        if launches > 1:
            out[label]["launches_per_call"] = launches
# tap_gather levels, labelled as bench.py names them (W = 256: Cout = 1)
for W, co in ((16, 512), (32, 256), (64, 128), (128, 64), (256, 1)):
    kf = [k for k in fetch if k[0].startswith("dmn::tap_gather_tile_kernel<%d," % W) and k in write]
    if not kf:
        continue
    alg = 13 * 32 * co * W * W * 4
    hb = int(round((2 * fetch[kf[0]] + write[kf[0]]) * 1024))
    out["tap_gather (9 planes -> up x2) %dx%d" % (W, W)] = {
        "fetch_size_kib": round(fetch[kf[0]], 1), "write_size_kib": round(write[kf[0]], 1), "hbm_bytes": hb,
        "algorithmic_bytes": alg, "ratio": round(hb / alg, 4)}
print(json.dumps(out, indent=1))
