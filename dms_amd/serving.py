# -*- coding: utf-8 -*-
"""Fixed-shape inference through ONE captured HIP graph.

At small batch the forward is ~700 short kernels and the host cannot issue them as fast as
the GPU finishes them (batch 1, 512 px on MI355X: 5.3 ms eager vs 4.0 ms of GPU work).
``GraphedDMN`` records one forward into a HIP graph (``torch.cuda.CUDAGraph`` is hipGraph on
ROCm; the ctypes launches of libdmn_hip.so go to torch's current stream, so they are
captured like any other kernel) and replays it per request.

The reference serves one (image, query) pair per forward (dmn_pytorch/models/dmn.py:142 is
batch-1 only); this is the same call, ``net(img, words)``, at a fixed shape.
"""
import torch


class GraphedDMN:
    """``g = GraphedDMN(net, batch, height, width, n_words); out = g(img, words)``.

    * ``net``: a ``dms_amd.DMN`` / ``BaseDMN`` in ``eval()`` mode on a ROCm device.
    * Inputs of exactly the captured shape; they are copied into the graph's static buffers.
    * The returned tensor is the graph's static output: it is overwritten by the next call
      (``clone()`` it to keep it).  Calls must not overlap on different streams.
    * Weights are read in place, but tensors DERIVED from them (folded comb_conv, BatchNorm
      constants, Winograd filter transforms) are frozen into the graph: call ``recapture()``
      after loading new weights.
    """

    def __init__(self, net, batch, height, width, n_words, warmup=3):
        if net.training:
            raise RuntimeError("GraphedDMN: put the network in eval() mode first")
        param = next(net.parameters())
        if not param.is_cuda:
            raise RuntimeError("GraphedDMN: the network must live on a ROCm device")
        self.net = net
        self.device = param.device
        self.warmup = int(warmup)
        self.img = torch.zeros((batch, 3, height, width), device=self.device)
        self.words = torch.zeros((batch, n_words), dtype=torch.long, device=self.device)
        self.graph = None
        self.out = None
        self.recapture()

    def recapture(self):
        with torch.no_grad(), torch.cuda.device(self.device):
            # warm-up on a side stream: MIOpen's find phase, the per-shape BLAS choices and the
            # weight-derived caches must all be settled before capture
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(self.warmup, 1)):
                    self.net(self.img, self.words)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                out = self.net(self.img, self.words)
            self.out = out[0] if isinstance(out, tuple) else out
        return self

    def __call__(self, img, words):
        if tuple(img.shape) != tuple(self.img.shape) or tuple(words.shape) != tuple(self.words.shape):
            raise RuntimeError("GraphedDMN: captured for img %s / words %s, got %s / %s"
                               % (tuple(self.img.shape), tuple(self.words.shape),
                                  tuple(img.shape), tuple(words.shape)))
        self.img.copy_(img, non_blocking=True)
        self.words.copy_(words, non_blocking=True)
        self.graph.replay()
        return self.out
