"""GPU test of dmn_sru_fwd_scan_noalloc (the one entry point that allocates nothing): its
chunk-carry workspace is a file-scope device array inside libdmn_hip.so, everything else is
dmn_sru_fwd_scan.  Checked through the C ABI against dmn_sru_fwd_scan with a caller workspace,
bit for bit, on two calls in a row (the second with a smaller input, so stale carries of the
first call sit in the scratch), plus the size limit.

Collected last on purpose (file name): the round-end run uses `pytest -x`.
"""
import pytest
import torch

from dms_amd import _lib, ops
from oracle.detgen import det_uniform

pytestmark = pytest.mark.gpu
DEV = "cuda"
SCRATCH_FLOATS = 1 << 20                      # DMN_SCAN_SCRATCH_FLOATS in include/dmn_hip.h


def _run_pair(T, B, d, k, nchunks, h_t0, seed, with_c0):
    u = det_uniform((T, B, d, k), seed, 1.5).to(DEV)
    x = det_uniform((T, B, d), seed + 1, 1.0).to(DEV) if k == 3 else None
    bias = det_uniform((2 * d,), seed + 2, 0.5).to(DEV)
    c0 = det_uniform((B, d), seed + 3, 1.0).to(DEV) if with_c0 else None
    # reference: the existing entry point with a caller-owned workspace
    h_ref, _, cl_ref = ops.sru_forward_raw(u, x, bias, c0, "tanh", 1.0, h_t0, want_c=False, nchunks=nchunks)
    # outputs are the caller's; poisoned so that an entry point that does nothing cannot pass
    h = torch.full((T - h_t0, B, d), float("nan"), device=DEV)
    cl = torch.full((B, d), float("nan"), device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.call("dmn_sru_fwd_scan_noalloc", u.data_ptr(), None if x is None else x.data_ptr(), bias.data_ptr(),
              None if c0 is None else c0.data_ptr(), h.data_ptr(), cl.data_ptr(),
              T, B, d, k, ops.ACT_IDS["tanh"], 1.0, h_t0, nchunks, stream)
    torch.cuda.synchronize()
    assert torch.equal(h, h_ref), "h differs from dmn_sru_fwd_scan"
    assert torch.equal(cl, cl_ref), "c_last differs from dmn_sru_fwd_scan"


def test_noalloc_scan_equals_workspace_scan_on_two_calls_in_a_row():
    _lib.load()
    # batch-1 serving shape of the mSRU (1005 columns, k = 4), 32 chunks ...
    _run_pair(T=1280, B=1, d=1005, k=4, nchunks=32, h_t0=640, seed=501, with_c0=False)
    # ... then a SMALLER input right behind it (fewer columns, fewer chunks, k = 3, an initial state)
    _run_pair(T=300, B=2, d=130, k=3, nchunks=7, h_t0=0, seed=511, with_c0=True)
    # and the first shape again: the scratch now holds the second call's carries
    _run_pair(T=1280, B=1, d=1005, k=4, nchunks=32, h_t0=640, seed=521, with_c0=True)


def test_noalloc_scan_does_not_allocate_device_memory():
    _lib.load()
    T, B, d, k, nch = 640, 1, 1005, 4, 16
    u = det_uniform((T, B, d, k), 531, 1.5).to(DEV)
    bias = det_uniform((2 * d,), 532, 0.5).to(DEV)
    h = torch.empty((T, B, d), device=DEV)
    cl = torch.empty((B, d), device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    args = (u.data_ptr(), None, bias.data_ptr(), None, h.data_ptr(), cl.data_ptr(), T, B, d, k, 1, 1.0, 0, nch, stream)
    _lib.call("dmn_sru_fwd_scan_noalloc", *args)            # first call on this device: symbol look-up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    reserved0 = torch.cuda.memory_reserved()
    for _ in range(3):
        _lib.call("dmn_sru_fwd_scan_noalloc", *args)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert torch.cuda.memory_reserved() == reserved0 and free1 == free0


def test_noalloc_scan_rejects_sizes_beyond_its_scratch():
    lib = _lib.load()
    # 2 * nchunks * B * d floats must fit DMN_SCAN_SCRATCH_FLOATS; rejected before any launch
    B, d, nch = 64, 1005, 16
    assert 2 * nch * B * d > SCRATCH_FLOATS
    rc = lib.dmn_sru_fwd_scan_noalloc(16, None, 16, None, 16, 16, 128, B, d, 4, 1, 1.0, 0, nch, None)
    assert rc == -11 and b"exceed the built-in scratch" in lib.dmn_last_error()
    with pytest.raises(RuntimeError, match="exceed the built-in scratch"):
        _lib.call("dmn_sru_fwd_scan_noalloc", 16, None, 16, None, 16, 16, 128, B, d, 4, 1, 1.0, 0, nch, None)
