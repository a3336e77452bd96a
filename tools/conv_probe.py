#!/usr/bin/env python
"""Where do the MIOpen convolutions stand? backbone / upsampling timings under layout options."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dms_amd
dev = "cuda"
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/iters
torch.manual_seed(0)
net = dms_amd.DMN(1000, num_filters=10, lang_layers=3, mix_we=True, pretrained=False, high_res=True, upsampling_mode="bilinear").to(dev).eval()
B = 32
img = torch.randn(B,3,512,512, device=dev)
with torch.no_grad():
    t = timeit(lambda: net.langvis.base(img)); print("backbone NCHW            : %.2f ms" % t, flush=True)
    vis, feats = net.langvis.base(img)
    x = torch.randn(B,1005,16,16, device=dev)
    t = timeit(lambda: net.upsample(x, feats)); print("upsample head NCHW       : %.2f ms" % t, flush=True)
    torch.backends.cudnn.benchmark = True
    t = timeit(lambda: net.langvis.base(img)); print("backbone NCHW benchmark  : %.2f ms" % t, flush=True)
    t = timeit(lambda: net.upsample(x, feats)); print("upsample NCHW benchmark  : %.2f ms" % t, flush=True)
    torch.backends.cudnn.benchmark = False
    base_cl = net.langvis.base.to(memory_format=torch.channels_last)
    img_cl = img.contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: base_cl(img_cl)); print("backbone channels_last   : %.2f ms" % t, flush=True)
