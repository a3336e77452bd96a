# -*- coding: utf-8 -*-
"""oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY; run ONCE in the build container.

    python -B oracle/gen_golden.py            # writes tests/golden/*.npz

Imports the reference's own ``dmn_pytorch/models/dmn.py`` from /root/reference (which
does not travel to the GPU box) and records its outputs on deterministic inputs
(oracle/detgen.py), so tests can pin oracle/dmn_ref.py -- and, through it, the HIP
path -- against what the reference itself computes.  Only data (inputs are rebuilt
from seeds; outputs are stored) lands in tests/golden/.

What is / is not reference code in those vectors:
* dmn.py's glue (BaseDMN.forward, UpsamplingModule.forward, DMN.forward,
  generate_spatial_batch) and the DPN92 definition run UNMODIFIED.
* The third-party `sru` package is absent (README.md:33), so ``sru.SRU`` is bound to
  oracle/sru_ref.SRURef -- the SRU arithmetic in these vectors is OUR restatement
  ("parity unpinned" for that part).
* Modules that dmn.py's import chain pulls in at import time but the dpn92 path never
  calls (torchvision.*) are bound to empty placeholder modules.
* ``Tensor.cuda`` is made a no-op for the hard-coded ``.cuda()`` at dmn.py:246.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import sru_ref                              # noqa: E402
from oracle.detgen import det_uniform, det_ints, det_fill_state   # noqa: E402


def import_reference():
    """Load dmn_pytorch.models.dmn without running dmn_pytorch/__init__.py
    (which would import the dataset loader -> cv2 / referit)."""
    sru_mod = types.ModuleType("sru")
    sru_mod.SRU = sru_ref.SRURef
    sys.modules["sru"] = sru_mod
    for name in ["torchvision", "torchvision.models", "torchvision.models.resnet",
                 "torchvision.models.densenet", "torchvision.models.inception",
                 "torchvision.models.vgg", "torchvision.transforms"]:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    for name, attrs in [("torchvision.models.resnet", ["resnet18", "resnet34", "resnet50",
                                                       "resnet101", "resnet152"]),
                        ("torchvision.models.densenet", ["densenet121", "densenet169",
                                                         "densenet161", "densenet201"]),
                        ("torchvision.models.inception", ["inception_v3"]),
                        ("torchvision.models.vgg", ["vgg16"])]:
        for a in attrs:
            setattr(sys.modules[name], a, None)
    pkg = types.ModuleType("dmn_pytorch")
    pkg.__path__ = [os.path.join(REF, "dmn_pytorch")]
    sys.modules["dmn_pytorch"] = pkg
    sys.dont_write_bytecode = True
    torch.Tensor.cuda = lambda self, *a, **k: self      # dmn.py:246
    return importlib.import_module("dmn_pytorch.models.dmn")


class FixedBackbone(nn.Module):
    """Stands where the DPN goes: returns pre-set (vis, features)."""

    def __init__(self, vis, feats):
        super().__init__()
        self.vis, self.feats = vis, feats

    def forward(self, x):
        return self.vis, list(self.feats)


def run_base(dmn, cfg, H, W, L, dict_size, tag, extra=None):
    vis = det_uniform((1, cfg["vis_size"], H, W), 101, 1.0).clamp_(min=0)   # post-ReLU like
    lang = det_ints((1, L), 202, dict_size)
    saved = dmn.create_model
    dmn.create_model = lambda *a, **k: FixedBackbone(vis, [])
    try:
        net = dmn.BaseDMN(dict_size, emb_size=cfg["emb_size"], hid_size=cfg["hid_size"],
                          vis_size=cfg["vis_size"], num_filters=cfg["num_filters"],
                          mixed_size=cfg["mixed_size"], hid_mixed_size=cfg["hid_mixed_size"],
                          lang_layers=cfg["lang_layers"], mixed_layers=cfg["mixed_layers"],
                          backend="dpn92", mix_we=cfg["mix_we"], pretrained=False,
                          high_res=cfg["high_res"])
    finally:
        dmn.create_model = saved
    net.eval()
    state = det_fill_state(net.state_dict())
    nn.Module.load_state_dict(net, state)
    with torch.no_grad():
        out, _ = net(torch.zeros(1, 3, H * 32, W * 32), lang)
    rec = {"out": out.numpy(), "H": H, "W": W, "L": L, "dict_size": dict_size,
           "keys": np.array(sorted(state.keys())),
           "shapes": np.array([str(tuple(state[k].shape)) for k in sorted(state.keys())])}
    for k, v in cfg.items():
        rec["cfg_" + k] = v
    if extra:
        rec.update(extra)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **rec)
    print(tag, "out", tuple(out.shape), float(out.abs().mean()), float(out.std()))
    return net


def main():
    os.makedirs(OUT, exist_ok=True)
    dmn = import_reference()

    # (4) coordinate maps, dmn.py:225-246
    rec = {}
    for (H, W) in [(10, 10), (16, 16), (16, 12), (1, 1), (3, 5)]:
        rec["s_%d_%d" % (H, W)] = dmn.BaseDMN.generate_spatial_batch(H, W).numpy()
    np.savez_compressed(os.path.join(OUT, "spatial.npz"), **rec)

    # (1) tiny-dims BaseDMN: k=3 everywhere (hid_mixed == mixed) and k=4 first layer
    tiny = dict(emb_size=16, hid_size=16, vis_size=32, num_filters=3, mixed_size=16,
                hid_mixed_size=16, lang_layers=2, mixed_layers=3, mix_we=True, high_res=False)
    run_base(dmn, tiny, 4, 3, 5, 50, "base_tiny_k3")
    run_base(dmn, dict(tiny, hid_mixed_size=21), 4, 3, 5, 50, "base_tiny_k4")
    run_base(dmn, dict(tiny, hid_mixed_size=21, mix_we=False, high_res=True, lang_layers=3),
             3, 4, 7, 50, "base_tiny_k4_nowe_hr")

    # (2) full-dims hot path (config 1's shapes after the backbone: 320px -> 10x10)
    full = dict(emb_size=1000, hid_size=1000, vis_size=2688, num_filters=10, mixed_size=1000,
                hid_mixed_size=1005, lang_layers=3, mixed_layers=3, mix_we=True, high_res=False)
    run_base(dmn, full, 10, 10, 20, 1000, "base_full_320_lowres")
    run_base(dmn, dict(full, high_res=True), 16, 16, 20, 1000, "base_full_512_highres_feat")

    # (5) UpsamplingModule: small channel counts, all five x2 levels; and one level with
    # a size mismatch that triggers the resize branch (dmn.py:332-336)
    fch = [6, 5, 4, 3, 2]
    up = dmn.UpsamplingModule(7, mode="bilinear", ker_size=3, amplification=8,
                              feature_channels=fch)
    up.eval()
    st = det_fill_state(up.state_dict())
    up.load_state_dict(st)
    x = det_uniform((1, 7, 3, 4), 303, 1.0)
    feats = [None, None,
             det_uniform((1, fch[2], 12, 16), 306, 1.0),
             det_uniform((1, fch[1], 6, 8), 305, 1.0),
             det_uniform((1, fch[0], 3, 4), 304, 1.0)]
    with torch.no_grad():
        y = up(x, feats)
    feats_mis = [None, None,
                 det_uniform((1, fch[2], 13, 15), 309, 1.0),
                 det_uniform((1, fch[1], 7, 9), 308, 1.0),
                 det_uniform((1, fch[0], 4, 5), 307, 1.0)]
    with torch.no_grad():
        y2 = up(x, feats_mis)
    np.savez_compressed(os.path.join(OUT, "upsample_small.npz"), out=y.numpy(), out_mis=y2.numpy(),
                        keys=np.array(sorted(st.keys())),
                        shapes=np.array([str(tuple(st[k].shape)) for k in sorted(st.keys())]))
    print("upsample_small", tuple(y.shape), tuple(y2.shape), float(y.std()))

    # (6) whole reference DMN incl. its own DPN92, 64px image (2x2 feature map), high-res:
    # pins state_dict keys/shapes of the real model and an end-to-end output
    torch.manual_seed(1111)
    net = dmn.DMN(100, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                  high_res=True, upsampling_mode="bilinear", upsampling_amplification=32)
    net.eval()
    sd = net.state_dict()
    keys = sorted(sd.keys())
    np.savez_compressed(os.path.join(OUT, "dmn_state_keys_highres.npz"),
                        keys=np.array(keys),
                        shapes=np.array([str(tuple(sd[k].shape)) for k in keys]))
    net_lo = dmn.DMN(100, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                     high_res=False)
    sdl = net_lo.state_dict()
    keys = sorted(sdl.keys())
    np.savez_compressed(os.path.join(OUT, "dmn_state_keys_lowres.npz"),
                        keys=np.array(keys),
                        shapes=np.array([str(tuple(sdl[k].shape)) for k in keys]))
    st = det_fill_state(sd)
    nn.Module.load_state_dict(net, st)
    img = det_uniform((1, 3, 64, 64), 404, 1.0)
    words = det_ints((1, 6), 405, 100)
    with torch.no_grad():
        y = net(img, words)
        vis, feats = net.langvis.base(img)
    np.savez_compressed(os.path.join(OUT, "dmn_e2e_64px_highres.npz"), out=y.numpy(),
                        vis=vis.numpy(), feat0=feats[0].numpy()[:, :, ::4, ::4],
                        feat3=feats[3].numpy())
    print("dmn_e2e_64px_highres", tuple(y.shape), float(y.std()), len(sd))


def gen_text_and_eval():
    """word_utils.Corpus (pure python + torch: imported as is) and train.py's
    compute_mask_IU (train.py executes argparse + dataset loading at import, so only that
    function's source is compiled, via ast) -> tests/golden/{tokenize,eval_iou}.json/.npz"""
    import ast
    import json
    import importlib.util
    # loaded by path: dmn_pytorch/utils/__init__.py would import transforms -> cv2 (absent)
    spec = importlib.util.spec_from_file_location(
        "ref_word_utils", os.path.join(REF, "dmn_pytorch", "utils", "word_utils.py"))
    wu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wu)
    vocab_lines = ["The left horse", "man in a red shirt standing", "second zebra from the right .",
                   "Bottle", "a CAT on the couch, sleeping", "person's hand", "giraffe"]
    queries = ["the left horse", "The man in the RED shirt.", "zebra, second from right .",
               "an unknown-word here", "  padded   spaces  ", "horse .", "cat",
               "a a a a a a a a a a a a a a a a a a a a a a a a the end", "person's hand!"]
    c = wu.Corpus()
    for l in vocab_lines:
        c.add_to_corpus(l)
    c.dictionary.add_word(wu.UNK_TOKEN)
    c.dictionary.add_word(wu.PAD_TOKEN)
    rec = {"vocab_lines": vocab_lines, "queries": queries, "idx2word": c.dictionary.idx2word,
           "ids_20": [c.tokenize(q, 20).tolist() for q in queries],
           "ids_5": [c.tokenize(q, 5).tolist() for q in queries],
           "ids_nolimit": [c.tokenize(q, -1).tolist() for q in queries]}
    with open(os.path.join(OUT, "tokenize.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("tokenize.json", len(queries), "queries, vocab", len(c))

    src = open(os.path.join(REF, "dmn_pytorch", "train.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "compute_mask_IU"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "train.py:compute_mask_IU", "exec"), ns)
    compute_mask_IU = ns["compute_mask_IU"]
    # the threshold sweep of evaluate(), train.py:343-374, on deterministic inputs
    score_thresh = np.concatenate([np.arange(start=0.00, stop=0.96, step=0.025)]).tolist()
    N, Hh, Ww = 3, 37, 29
    prob = torch.sigmoid(det_uniform((N, Hh, Ww), 501, 4.0))
    prob[0, 0, :5] = torch.tensor(score_thresh[10:15], dtype=torch.float32)     # exact ties
    mask = (det_uniform((N, Hh, Ww), 502, 1.0) > 0.2).float()
    mask[2] = 0                                                               # empty target
    inter = torch.zeros(N, len(score_thresh))
    union = torch.zeros(N, len(score_thresh))
    for n in range(N):
        for idx, thresh in enumerate(score_thresh):
            inter[n, idx], union[n, idx] = compute_mask_IU((prob[n] > thresh).float(), mask[n])
    np.savez_compressed(os.path.join(OUT, "eval_iou.npz"), inter=inter.numpy(), union=union.numpy(),
                        thresholds=np.array(score_thresh, dtype=np.float64), N=N, H=Hh, W=Ww)
    print("eval_iou.npz", inter.shape, float(inter.sum()), float(union.sum()))


def gen_data_files():
    """The reference's on-disk query formats (referit_loader.py:134-138,153-157,194-199), written by
    the reference's OWN classes: a word_utils.Corpus built by Corpus.load_file from a vocabulary
    file and torch.save'd as corpus.pth (pickled under its real module path), and the split lists
    of (image, mask, phrase) triples -> tests/golden/refdata/<dataset>/ (+ tokenize_phrase ids)."""
    import importlib.util
    import json
    name = "dmn_pytorch.utils.word_utils"                     # the path a real corpus.pth is pickled under
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "dmn_pytorch", "utils", "word_utils.py"))
    wu = importlib.util.module_from_spec(spec)
    for parent in ("dmn_pytorch", "dmn_pytorch.utils"):       # empty packages: their __init__ imports cv2
        pkg = types.ModuleType(parent)
        pkg.__path__ = []
        sys.modules.setdefault(parent, pkg)
    sys.modules[name] = wu
    spec.loader.exec_module(wu)
    root = os.path.join(OUT, "refdata")
    vocab = ["the", "left", "horse", "man", "in", "a", "red", "shirt", "standing", "second", "zebra",
             "from", "right", "bottle", "cat", "on", "couch", "sleeping", "person", "hand", "giraffe",
             "woman", "blue", "umbrella", "dog", "near", "tree", "big", "small", "table"]
    phrases = {"train": ["the left horse", "man in a red shirt standing", "second zebra from the right .",
                         "bottle", "a cat on the couch , sleeping"],
               "val": ["woman in blue", "the big dog near the tree", "small table", "giraffe",
                       "person hand on the left of the red umbrella near the big tree in a blue shirt standing on a table",
                       "an unseen-word , here !"]}
    rec = {}
    for dataset in ("unc", "referit"):
        d = os.path.join(root, dataset)
        os.makedirs(d, exist_ok=True)
        vf = os.path.join(d, "vocabulary.txt")
        with open(vf, "w") as f:
            f.write("\n".join(vocab) + "\n")
        c = wu.Corpus()
        c.load_file(vf)                                        # adds <unk>, <pad> (word_utils.py:53-60)
        os.remove(vf)
        torch.save(c, os.path.join(d, "corpus.pth"))
        # PyTorch 0.4 (the reference's era) wrote the legacy, non-zip format
        torch.save(c, os.path.join(d, "corpus_legacy.pth"), _use_new_zipfile_serialization=False)
        for split, ps in phrases.items():
            triples = [("COCO_train2014_%012d.jpg" % (7 * i + 1), "%d.pth" % (100 + i), p) for i, p in enumerate(ps)]
            torch.save(triples, os.path.join(d, "%s_%s.pth" % (dataset, split)))
            rec["%s_%s" % (dataset, split)] = {"triples": [list(t) for t in triples],
                                               "ids_20": [c.tokenize(p, 20).tolist() for p in ps],
                                               "ids_8": [c.tokenize(p, 8).tolist() for p in ps]}
        rec[dataset + "_idx2word"] = c.dictionary.idx2word
    with open(os.path.join(root, "expected.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("refdata:", sorted(os.listdir(os.path.join(root, "unc"))), "vocab", len(c))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "datafiles":
        gen_data_files()
    elif len(sys.argv) > 1 and sys.argv[1] == "extras":
        import_reference()
        gen_text_and_eval()
    else:
        main()
        gen_text_and_eval()
