import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from dms_amd import ops
from oracle import sru_ref
from oracle.detgen import det_uniform
def case(T,B,d,k,seed):
    u = det_uniform((T,B,d,k), seed, 1.5); x = det_uniform((T,B,d), seed+1, 1.0) if k==3 else None
    bias = det_uniform((2*d,), seed+2, 0.5); c0 = det_uniform((B,d), seed+3, 1.0)
    return u,x,bias,c0
for (T,B,d,k) in [(64,1,64,3),(64,1,64,4),(40,1,64,3),(100,1,64,3)]:
  for depth in (4,8,16,32):
    for want_c in (True, False):
        os.environ["DMN_SRU_DEPTH"]=str(depth)
        u,x,bias,c0 = case(T,B,d,k,41)
        h_ref,c_ref,cl = sru_ref.sru_fwd_c(u.numpy(), None if x is None else x.numpy(), bias.numpy(), c0.numpy(), "tanh", 1.0)
        h,c_all,c_last = ops.sru_forward_raw(u.cuda(), None if x is None else x.cuda(), bias.cuda(), c0.cuda(), "tanh", 1.0, 0, want_c=want_c, nchunks=1)
        err = np.abs(h.cpu().numpy()-h_ref).max(axis=(1,2))
        bad = np.nonzero(err>1e-4)[0]
        print(T,B,d,k,"depth",depth,"want_c",want_c,"bad rows",bad.tolist()[:20], flush=True)
