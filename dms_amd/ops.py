# -*- coding: utf-8 -*-
"""Torch-facing wrappers of the HIP entry points (include/dmn_hip.h).

PyTorch is plumbing here: it owns device memory, streams and autograd bookkeeping; every
operator below runs in libdmn_hip.so.  Tensors must be fp32 and on a ROCm device --
anything else raises (there is no CPU or eager fallback).
"""
import contextlib
import os
import tempfile

import torch

from . import _lib

ACT_IDS = {"identity": 0, "tanh": 1, "relu": 2}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(name, t, shape=None):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("dms_amd.%s: tensor is on %s; the HIP path needs a ROCm device "
                           "(no CPU fallback exists)" % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("dms_amd.%s: expected float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("dms_amd.%s: tensor must be contiguous" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError("dms_amd.%s: expected shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _chk_batch_strided(name, t, shape=None):
    """Like _chk, but the batch dimension may be strided (a channel-slice view of a larger
    (N, C', H, W) buffer): every sample must be contiguous.  Returns the batch stride in floats."""
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError("dms_amd.%s: expected a float32 ROCm tensor" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError("dms_amd.%s: expected shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    if not t[0].is_contiguous() or t.stride(0) < t[0].numel():
        raise RuntimeError("dms_amd.%s: every sample must be contiguous" % name)
    return int(t.stride(0))


_BLAS_NAMES = {"rocblas": "cublas", "hipblaslt": "cublaslt"}


@contextlib.contextmanager
def gemm_backend():
    """Route the hot path's plain fp32 library GEMMs (x.W projections, filter Linear) to the
    BLAS library that is faster for their shapes on MI355X: rocBLAS, measured 142.6 vs
    128.8 TFLOP/s (hipBLASLt, torch's default here) on the mSRU projection
    163840x1005 . 1005x3015 (profiles/r1_gemm_backend_probe.txt).  DMN_GEMM_BACKEND =
    rocblas | hipblaslt | default overrides.  The preference is process-global in torch, so
    it is restored on exit; not safe against other threads issuing GEMMs concurrently."""
    want = os.environ.get("DMN_GEMM_BACKEND", "rocblas")
    if want == "default":
        yield
        return
    if want not in _BLAS_NAMES:
        raise RuntimeError("DMN_GEMM_BACKEND must be rocblas, hipblaslt or default (got %r)" % want)
    prev = torch.backends.cuda.preferred_blas_library()
    torch.backends.cuda.preferred_blas_library(_BLAS_NAMES[want])
    try:
        yield
    finally:
        torch.backends.cuda.preferred_blas_library(prev)


GEMM_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "gemm_mi355x_512px.csv")


GEMM_TABLE_TRAIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "gemm_mi355x_train320.csv")


def use_gemm_table(path=None):
    """Opt in to recorded per-shape solution tables for the library GEMMs (PyTorch TunableOp in
    look-up mode: no tuning at run time).  The tables shipped in dms_amd/tuning/ were searched on
    one MI355X over all rocBLAS + hipBLASLt fp32 solutions for every GEMM shape of the B=32 and
    the B=1 512 px inference step (tools/tune_gemms.py) and of the B=64 320 px training step
    (forward, dgrad, wgrad; tools/tune_train_gemms.py); shapes they do not hold keep the library's
    own choice.  TunableOp ignores a file unless its validator lines (torch, HIP, rocBLAS,
    hipBLASLt versions, gfx arch) match the running stack.  Process-global.  `path`: one file, or
    several separated by os.pathsep (also DMN_GEMM_TABLE).  Returns the number of entries loaded
    (0: files missing or rejected)."""
    import torch.cuda.tunable as tunable
    path = path or os.environ.get("DMN_GEMM_TABLE", GEMM_TABLE + os.pathsep + GEMM_TABLE_TRAIN)
    if path in ("0", "off"):
        return 0
    files = [p for p in path.split(os.pathsep) if p and os.path.exists(p)]
    if not files:
        return 0
    tunable.enable(True)
    tunable.tuning_enable(False)
    ok = False
    for f in files:
        ok = bool(tunable.read_file(f)) or ok
    # TunableOp writes its results back to "its" file; keep that away from the shipped tables
    # (several ranks share them) by pointing it at a per-process scratch name
    tunable.set_filename(os.path.join(tempfile.gettempdir(), "dmn_tunableop_unused_%d.csv" % os.getpid()))
    n = len(tunable.get_results()) if ok else 0
    if not n:
        tunable.enable(False)
    return n


_BMM_CHOICE = {}


def _time_matmul(name, a, b, reps=3):
    torch.backends.cuda.preferred_blas_library(name)
    torch.matmul(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.matmul(a, b)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1)


def tuned_matmul(a, b, acc=None, out=None):
    """torch.matmul(a, b) (+ acc, accumulated in place, when given: a (M, K), b (B, K, N),
    acc (B, M, N); or written into `out` (B, M, N), which like acc may be a row-slice view of a
    larger (B, rows, N) buffer: unit stride along N) with the BLAS library (rocBLAS / hipBLASLt) chosen PER SHAPE: neither
    wins everywhere on MI355X (profiles/r1_conv1x1_backend_probe.txt: hipBLASLt is 10-25 % faster
    on DPN92's conv2_x / conv4_x 1x1 shapes, rocBLAS on the others and on the mSRU projections).
    The choice is timed once per shape on first use (a few launches and one event wait; not
    during stream capture, where rocBLAS is used) and cached for the process, so results are
    reproducible within a process and may differ in the last bits between processes.
    DMN_GEMM_TUNE=0 disables the timing (always rocBLAS).  Inference only (operands detached:
    torch.matmul folds -- and copies -- a broadcast operand that requires grad)."""
    a, b = a.detach(), b.detach()
    key = (tuple(a.shape), tuple(b.shape), a.device.index)
    prev = torch.backends.cuda.preferred_blas_library()
    try:
        choice = _BMM_CHOICE.get(key)
        if choice is None:
            if os.environ.get("DMN_GEMM_TUNE", "1") == "0" or torch.cuda.is_current_stream_capturing():
                choice = "cublas"
            else:
                t = {name: _time_matmul(name, a, b) for name in ("cublas", "cublaslt")}
                choice = min(t, key=t.get)
                _BMM_CHOICE[key] = choice
        torch.backends.cuda.preferred_blas_library(choice)
        if acc is not None:
            out = torch.baddbmm(acc, a.unsqueeze(0).expand(b.shape[0], -1, -1), b, out=acc)
        elif out is not None:
            out = torch.baddbmm(out, a.unsqueeze(0).expand(b.shape[0], -1, -1), b, beta=0, out=out)
        else:
            out = torch.matmul(a, b)
    finally:
        torch.backends.cuda.preferred_blas_library(prev)
    return out


def conv1x1(x, weight):
    """1x1 / stride-1 convolution of an NCHW tensor as the batched library GEMM
    W (Cout x Cin) . X[n] (Cin x HW) -- what MIOpen itself runs for this case (its result is
    bit-identical to the rocBLAS call) -- through tuned_matmul."""
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    _chk("conv1x1", x)
    out = tuned_matmul(weight.view(Cout, Cin), x.view(N, Cin, H * W))
    return out.view(N, Cout, H, W)


# ------------------------------------------------------------------------------------
# SRU scan
# ------------------------------------------------------------------------------------
def sru_scan_chunks(T, ncols):
    """How many time chunks the time-parallel variant should use (1 = plain column scan).

    Measured on MI355X (tools/scan_bench.py, T=5120, d=1005): from ~126 wavefronts (B=8) up
    the plain column scan wins (each wave already owns a SIMD and the chunked variant reads U
    twice); below that, ~512 wavefront-chunks in flight is the sweet spot (B=1: 291 us plain
    vs 39 us with 32 chunks).
    """
    waves = (ncols + 63) // 64
    if waves >= 100 or T < 128:
        return 1
    return max(min(512 // waves, T // 64), 1)


def sru_forward_raw(u, x, bias, c0, act, scale_x, h_t0=0, want_c=False, nchunks=None):
    """u (T,B,d,k); x (T,B,d) or None; returns h (T-h_t0,B,d), c_all|None, c_last (B,d)."""
    T, B, d, k = u.shape
    _chk("sru_fwd", u)
    _chk("sru_fwd", x, (T, B, d) if x is not None else None)
    _chk("sru_fwd", bias, (2 * d,))
    _chk("sru_fwd", c0, (B, d) if c0 is not None else None)
    with torch.cuda.device(u.device):
        h = torch.empty((T - h_t0, B, d), device=u.device, dtype=torch.float32)
        c_last = torch.empty((B, d), device=u.device, dtype=torch.float32)
        c_all = torch.empty((T, B, d), device=u.device, dtype=torch.float32) if want_c else None
        if nchunks is None:
            nchunks = 1 if want_c else sru_scan_chunks(T, B * d)
        if nchunks > 1 and not want_c:
            ws = torch.empty(2 * nchunks * B * d, device=u.device, dtype=torch.float32)
            _lib.call("dmn_sru_fwd_scan", _ptr(u), _ptr(x), _ptr(bias), _ptr(c0), _ptr(h),
                      _ptr(c_last), _ptr(ws), T, B, d, k, ACT_IDS[act], float(scale_x),
                      int(h_t0), int(nchunks), _stream())
        else:
            _lib.call("dmn_sru_fwd", _ptr(u), _ptr(x), _ptr(bias), _ptr(c0), _ptr(h),
                      _ptr(c_all), _ptr(c_last), T, B, d, k, ACT_IDS[act], float(scale_x),
                      int(h_t0), _stream())
    return h, c_all, c_last


def sru_forward_image(u, x, bias, c0, act, scale_x, out=None, nchunks=None):
    """The scan with h stored as an image (inference): u (T,B,d,k), x (T,B,d) or None -> out (B, d, T) with
    out[b, j, t] = h[t, b, j] -- what dmn.py:209-212 builds with permute(1, 2, 0) -- and c_last (B, d).
    `out` may be given: a batch-strided (B, d, T) view, e.g. the first d channels of the head's
    [x | feat] buffer.  Bit-identical to sru_forward_raw(...)[0].permute(1, 2, 0)."""
    T, B, d, k = u.shape
    _chk("sru_fwd_image", u)
    _chk("sru_fwd_image", x, (T, B, d) if x is not None else None)
    _chk("sru_fwd_image", bias, (2 * d,))
    _chk("sru_fwd_image", c0, (B, d) if c0 is not None else None)
    with torch.cuda.device(u.device):
        if out is None:
            out = torch.empty((B, d, T), device=u.device, dtype=torch.float32)
        bs = _chk_batch_strided("sru_fwd_image", out, (B, d, T))
        c_last = torch.empty((B, d), device=u.device, dtype=torch.float32)
        if nchunks is None:
            nchunks = sru_scan_chunks(T, B * d)
        ws = torch.empty(2 * nchunks * B * d, device=u.device, dtype=torch.float32) if nchunks > 1 else None
        _lib.call("dmn_sru_fwd_image", _ptr(u), _ptr(x), _ptr(bias), _ptr(c0), _ptr(out), _ptr(c_last),
                  _ptr(ws), T, B, d, k, ACT_IDS[act], float(scale_x), bs, int(nchunks), _stream())
    return out, c_last


def sru_state(u2, bias, c0=None, nchunks=None):
    """Cell state after the last step for u2 (T, B, d, 2) = [u0 | forget]; inference only.
    Few columns (small B): time is cut into chunks that run concurrently (sru_scan_chunks)."""
    T, B, d, two = u2.shape
    if two != 2:
        raise RuntimeError("dms_amd.sru_state: expected (T, B, d, 2), got %s" % (tuple(u2.shape),))
    _chk("sru_state", u2)
    _chk("sru_state", bias, (2 * d,))
    _chk("sru_state", c0, (B, d) if c0 is not None else None)
    with torch.cuda.device(u2.device):
        c_last = torch.empty((B, d), device=u2.device, dtype=torch.float32)
        if nchunks is None:
            nchunks = sru_scan_chunks(T, B * d)
        if nchunks > 1:
            ws = torch.empty(2 * nchunks * B * d, device=u2.device, dtype=torch.float32)
            _lib.call("dmn_sru_state_fwd_scan", _ptr(u2), _ptr(bias), _ptr(c0), _ptr(c_last), _ptr(ws),
                      T, B, d, int(nchunks), _stream())
        else:
            _lib.call("dmn_sru_state_fwd", _ptr(u2), _ptr(bias), _ptr(c0), _ptr(c_last), T, B, d, _stream())
    return c_last


def _bias_partials(n, device):
    """Scratch for the deterministic batch reduction of the bias gradient (sample order, as the reference sums its
    (2, B, d) partials): the default.  DMN_SRU_BWD_ATOMIC=1 -> None: in-kernel float atomics, order not fixed."""
    if os.environ.get("DMN_SRU_BWD_ATOMIC") == "1":
        return None
    return torch.empty((n,), device=device, dtype=torch.float32)


class _SRUStateCompute(torch.autograd.Function):
    """Autograd node of the state-only scan: c_last from u2 = [u0 | forget] (training: the steps of the
    last mSRU layer whose hidden state nothing reads, so only the cell-state chain carries gradient)."""

    @staticmethod
    def forward(ctx, u2, bias, c0):
        T, B, d, _ = u2.shape
        with torch.cuda.device(u2.device):
            c_all = torch.empty((T, B, d), device=u2.device, dtype=torch.float32)
            c_last = torch.empty((B, d), device=u2.device, dtype=torch.float32)
            _lib.call("dmn_sru_state_fwd_train", _ptr(u2), _ptr(bias), _ptr(c0), _ptr(c_all), _ptr(c_last),
                      T, B, d, _stream())
        ctx.save_for_backward(u2, bias, c0, c_all)
        return c_last

    @staticmethod
    def backward(ctx, grad_last):
        u2, bias, c0, c_all = ctx.saved_tensors
        T, B, d, _ = u2.shape
        grad_last = grad_last.contiguous()
        with torch.cuda.device(u2.device):
            gu2 = torch.empty_like(u2)
            gbf = torch.empty((d,), device=u2.device, dtype=torch.float32)
            gc0 = torch.empty((B, d), device=u2.device, dtype=torch.float32)
            part = _bias_partials(B * d, u2.device)
            _lib.call("dmn_sru_state_bwd", _ptr(u2), _ptr(bias), _ptr(c0), _ptr(c_all), _ptr(grad_last),
                      _ptr(gu2), _ptr(gbf), _ptr(gc0), _ptr(part), T, B, d, _stream())
        gbias = torch.cat([gbf, torch.zeros_like(gbf)]) if ctx.needs_input_grad[1] else None
        return gu2, gbias, (gc0 if c0 is not None and ctx.needs_input_grad[2] else None)


def sru_state_train(u2, bias, c0=None):
    """sru_state under autograd: c_last (B, d) with gradients to u2 (T, B, d, 2), bias (2d) and c0."""
    T, B, d, two = u2.shape
    if two != 2:
        raise RuntimeError("dms_amd.sru_state_train: expected (T, B, d, 2), got %s" % (tuple(u2.shape),))
    u2 = u2.contiguous()
    _chk("sru_state_train", u2)
    _chk("sru_state_train", bias, (2 * d,))
    _chk("sru_state_train", c0, (B, d) if c0 is not None else None)
    return _SRUStateCompute.apply(u2, bias, c0.contiguous() if c0 is not None else None)


class _SRUCompute(torch.autograd.Function):
    """Autograd node of the scan (the role of upstream's SRU_Compute)."""

    @staticmethod
    def forward(ctx, u, x, bias, c0, act, scale_x):
        h, c_all, c_last = sru_forward_raw(u, x, bias, c0, act, scale_x, 0, want_c=True)
        ctx.save_for_backward(u, x, bias, c0, c_all)
        ctx.act, ctx.scale_x = act, scale_x
        return h, c_last

    @staticmethod
    def backward(ctx, grad_h, grad_last):
        u, x, bias, c0, c_all = ctx.saved_tensors
        T, B, d, k = u.shape
        grad_h = grad_h.contiguous()
        grad_last = grad_last.contiguous() if grad_last is not None else None
        with torch.cuda.device(u.device):
            gu = torch.empty_like(u)
            gx = torch.empty((T, B, d), device=u.device, dtype=torch.float32) if k == 3 else None
            gbias = torch.empty((2 * d,), device=u.device, dtype=torch.float32)   # summed over t and b by the library
            gc0 = torch.empty((B, d), device=u.device, dtype=torch.float32)
            part = _bias_partials(2 * B * d, u.device)
            _lib.call("dmn_sru_bwd", _ptr(u), _ptr(x), _ptr(bias), _ptr(c0), _ptr(c_all),
                      _ptr(grad_h), _ptr(grad_last), _ptr(gu), _ptr(gx), _ptr(gbias), _ptr(gc0), _ptr(part),
                      T, B, d, k, ACT_IDS[ctx.act], float(ctx.scale_x), _stream())
        return gu, gx, gbias, (gc0 if c0 is not None else None), None, None


def sru_compute(u, x, bias, c0=None, act="tanh", scale_x=1.0, h_t0=0):
    """h, c_last = scan(u, x).  With autograd: full h is produced, then sliced."""
    needs_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (u, x, bias, c0))
    if needs_grad:
        h, c_last = _SRUCompute.apply(u, x, bias, c0, act, scale_x)
        return (h[h_t0:] if h_t0 else h), c_last
    h, _, c_last = sru_forward_raw(u, x, bias, c0, act, scale_x, h_t0, want_c=False)
    return h, c_last


# ------------------------------------------------------------------------------------
# coordinate channels + concat
# ------------------------------------------------------------------------------------
class _CatCoords(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vis):
        B, C, H, W = vis.shape
        _chk("cat_coords", vis)
        with torch.cuda.device(vis.device):
            V = torch.empty((B, C + 8, H, W), device=vis.device, dtype=torch.float32)
            _lib.call("dmn_cat_coords", _ptr(vis), _ptr(V), B, C, H, W, _stream())
        ctx.C = C
        return V

    @staticmethod
    def backward(ctx, gV):
        return gV[:, :ctx.C].contiguous()


def cat_coords(vis):
    return _CatCoords.apply(vis.contiguous())


def bias_sigmoid_(x, bias):
    """x (rows, cols) <- sigmoid(x + bias[None, :]) in place (inference): the filter Linear's bias add and
    F.sigmoid (dmn.py:173-174) as one streaming pass behind a bias-free GEMM."""
    rows, cols = x.shape
    _chk("bias_sigmoid", x)
    _chk("bias_sigmoid", bias, (cols,))
    with torch.cuda.device(x.device):
        _lib.call("dmn_bias_sigmoid", _ptr(x), _ptr(bias), int(rows), int(cols), _stream())
    return x


# ------------------------------------------------------------------------------------
# dynamic filters
# ------------------------------------------------------------------------------------
class _DynFilter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, V, F):
        B, C, S = V.shape
        L, B2, K, C2 = F.shape
        if B2 != B or C2 != C:
            raise RuntimeError("dms_amd.dynfilter: V %s vs F %s" % (tuple(V.shape), tuple(F.shape)))
        _chk("dynfilter", V)
        _chk("dynfilter", F)
        with torch.cuda.device(V.device):
            P = torch.empty((L, B, K, S), device=V.device, dtype=torch.float32)
            # few samples: a workspace lets the library spread the contraction over more workgroups
            # (same bits either way); 16 tiles per sample at 512 px, so below ~64 samples
            nws = _lib.query("dmn_dynfilter_workspace_floats", B, L, K, C, S) if B * ((L * K + 63) // 64) * ((S + 63) // 64) < 1024 else 0
            if nws > 0:
                ws = torch.empty((nws,), device=V.device, dtype=torch.float32)
                _lib.call("dmn_dynfilter_fwd_ws", _ptr(V), _ptr(F), _ptr(P), _ptr(ws), B, L, K, C, S, _stream())
            else:
                _lib.call("dmn_dynfilter_fwd", _ptr(V), _ptr(F), _ptr(P), B, L, K, C, S, _stream())
        ctx.save_for_backward(V, F)
        return P

    @staticmethod
    def backward(ctx, gP):
        V, F = ctx.saved_tensors
        B, C, S = V.shape
        L, _, K, _ = F.shape
        gP = gP.contiguous()
        with torch.cuda.device(V.device):
            gV = torch.empty_like(V) if ctx.needs_input_grad[0] else None
            gF = torch.empty_like(F) if ctx.needs_input_grad[1] else None
            _lib.call("dmn_dynfilter_bwd", _ptr(V), _ptr(F), _ptr(gP), _ptr(gV), _ptr(gF),
                      B, L, K, C, S, _stream())
        return gV, gF


def dynfilter(V, F):
    """V (B,C,S), F (L,B,K,C) -> P (L,B,K,S);  P[t,b,k,s] = sum_c F[t,b,k,c] V[b,c,s]."""
    return _DynFilter.apply(V.contiguous(), F.contiguous())


# ------------------------------------------------------------------------------------
# concat-free mix
# ------------------------------------------------------------------------------------
class _MixBcastAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, Bt, P, Wp):
        B, S, M = A.shape
        L, _, K, _ = P.shape
        _chk("mix", A)
        _chk("mix", Bt, (L, B, M))
        _chk("mix", P, (L, B, K, S))
        if not (Wp.is_cuda and Wp.dtype == torch.float32 and tuple(Wp.shape) == (M, K)
                and Wp.stride(1) == 1):
            raise RuntimeError("dms_amd.mix: Wp must be a cuda float32 (M,K) view with unit inner stride")
        ldw = Wp.stride(0)
        with torch.cuda.device(A.device):
            Q = torch.empty((S * L, B, M), device=A.device, dtype=torch.float32)
            _lib.call("dmn_mix_bcast_add", _ptr(A), _ptr(Bt), _ptr(P), _ptr(Wp), ldw, _ptr(Q),
                      B, L, K, S, M, _stream())
        ctx.save_for_backward(P, Wp)
        return Q

    @staticmethod
    def backward(ctx, gQ):
        P, Wp = ctx.saved_tensors
        L, B, K, S = P.shape
        M = Wp.shape[0]
        gQ = gQ.contiguous()
        dev = gQ.device
        with torch.cuda.device(dev):
            gA = torch.empty((B, S, M), device=dev, dtype=torch.float32)
            gBt = torch.empty((L, B, M), device=dev, dtype=torch.float32)
            gP = torch.empty((L, B, K, S), device=dev, dtype=torch.float32)
            # gWp leaves the library as `parts` (M, K) slabs (one per sample and word group, or per sample
            # and word); wide rows at training batch sizes take the one-pass kernel, which wants a workspace
            parts = _lib.query("dmn_mix_bwd_parts", B, L, K, S, M)
            nws = _lib.query("dmn_mix_bwd_workspace_floats", B, L, K, S, M)
            gWpp = torch.empty((parts, M, K), device=dev, dtype=torch.float32)
            ws = torch.empty((nws,), device=dev, dtype=torch.float32) if nws > 0 else None
            _lib.call("dmn_mix_bcast_add_bwd", _ptr(gQ), _ptr(P), _ptr(Wp), Wp.stride(0),
                      _ptr(gA), _ptr(gBt), _ptr(gP), _ptr(gWpp), _ptr(ws), B, L, K, S, M, _stream())
        return gA, gBt, gP, gWpp.sum(dim=0)


def mix_bcast_add(A, Bt, P, Wp):
    """Q[(s*L+t), b, m] = A[b,s,m] + Bt[t,b,m] + sum_k Wp[m,k] P[t,b,k,s] -> (S*L, B, M)."""
    return _MixBcastAdd.apply(A.contiguous(), Bt.contiguous(), P.contiguous(), Wp)


# ------------------------------------------------------------------------------------
# cat + bilinear x2
# ------------------------------------------------------------------------------------
def mix_sru_fwd(A, Bt, P, Wp, bias, c0=None, act="tanh"):
    """mix_bcast_add fused into the layer-0 scan (inference, folded comb_conv, k = 4):
    A (B, S, 4d), Bt (L, B, 4d), P (L, B, K, S), Wp (4d, K) (a column slice is fine), bias (2d)
    -> h (S*L, B, d), c_last (B, d); bit-identical to sru_forward_raw(mix_bcast_add(...))."""
    B, S, M4 = A.shape
    L, _, K, _ = P.shape
    d = M4 // 4
    _chk("mix_sru_fwd", A, (B, S, 4 * d))
    _chk("mix_sru_fwd", Bt, (L, B, 4 * d))
    _chk("mix_sru_fwd", P, (L, B, K, S))
    _chk("mix_sru_fwd", bias, (2 * d,))
    _chk("mix_sru_fwd", c0, (B, d) if c0 is not None else None)
    if Wp.shape != (4 * d, K) or Wp.stride(1) != 1 or not Wp.is_cuda or Wp.dtype != torch.float32:
        raise RuntimeError("dms_amd.mix_sru_fwd: Wp must be a float32 ROCm (4d, K) matrix with unit column stride")
    with torch.cuda.device(A.device):
        Pt = P.permute(1, 3, 0, 2).contiguous()                   # (B, S, L, K): a pixel's L*K values adjacent
        h = torch.empty((S * L, B, d), device=A.device, dtype=torch.float32)
        c_last = torch.empty((B, d), device=A.device, dtype=torch.float32)
        _lib.call("dmn_mix_sru_fwd", _ptr(A), _ptr(Bt), _ptr(Pt), _ptr(Wp), int(Wp.stride(0)), _ptr(bias),
                  _ptr(c0), _ptr(h), _ptr(c_last), B, L, K, S, d, ACT_IDS[act], _stream())
    return h, c_last


class _CatUpsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, feat):
        B, Cx, H, W = x.shape
        B2, Cf, H2, W2 = feat.shape
        if (B2, H2, W2) != (B, H, W):
            raise RuntimeError("dms_amd.cat_upsample2x: x %s vs feat %s" % (tuple(x.shape), tuple(feat.shape)))
        _chk("cat_upsample2x", x)
        _chk("cat_upsample2x", feat)
        with torch.cuda.device(x.device):
            out = torch.empty((B, Cx + Cf, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
            _lib.call("dmn_cat_upsample2x", _ptr(x), _ptr(feat), _ptr(out), B, Cx, Cf, H, W, _stream())
        ctx.dims = (B, Cx, Cf, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Cx, Cf, H, W = ctx.dims
        g = g.contiguous()
        with torch.cuda.device(g.device):
            gx = torch.empty((B, Cx, H, W), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
            gf = torch.empty((B, Cf, H, W), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
            if gx is not None or gf is not None:
                _lib.call("dmn_cat_upsample2x_bwd", _ptr(g), _ptr(gx), _ptr(gf), B, Cx, Cf, H, W, _stream())
        return gx, gf


def cat_upsample2x(x, feat):
    """bilinear x2 (align_corners=False) of cat([x, feat], 1), one pass."""
    return _CatUpsample2x.apply(x.contiguous(), feat.contiguous())


# ------------------------------------------------------------------------------------
# Winograd F(2x2, 3x3) halves of a head level (inference only: no autograd)
# ------------------------------------------------------------------------------------
_WINO_G = {
    2: ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0)),
    4: ((1 / 4, 0.0, 0.0), (-1 / 6, -1 / 6, -1 / 6), (-1 / 6, 1 / 6, -1 / 6),
        (1 / 24, 1 / 12, 1 / 6), (1 / 24, -1 / 12, 1 / 6), (0.0, 0.0, 1.0)),
}


def wino_filter_transform(weight, m=2):
    """U ((m+2)^2, Cout, C) = G g G^T of a (Cout, C, 3, 3) weight (tiny; plain torch, fp64 inside)."""
    G = torch.tensor(_WINO_G[m], dtype=torch.float64, device=weight.device)
    Cout, C = weight.shape[:2]
    U = torch.einsum("ij,ocjk,lk->iloc", G, weight.double(), G)
    return U.reshape((m + 2) ** 2, Cout, C).to(weight.dtype).contiguous()


def wino_cat_upsample2x_transform(x, feat, m=2):
    """V ((m+2)^2, Cx+Cf, tiles): input transform of conv3x3(pad 1) applied to
    upsample2x_bilinear(cat([x, feat], 1)), straight from the low-res tensors.
    tiles = B * (2H/m) * (2W/m); m = 4 needs even H, W."""
    B, Cx, H, W = x.shape
    Cf = feat.shape[1]
    _chk("wino_cat_upsample2x_transform", x)
    _chk("wino_cat_upsample2x_transform", feat, (B, Cf, H, W))
    tiles = B * (2 * H // m) * (2 * W // m)
    with torch.cuda.device(x.device):
        V = torch.empty(((m + 2) ** 2, Cx + Cf, tiles), device=x.device, dtype=torch.float32)
        _lib.call("dmn_wino_cat_upsample2x_transform", _ptr(x), _ptr(feat), _ptr(V), B, Cx, Cf, H, W,
                  int(m), _stream())
    return V


def wino_output_transform(M, bias, B, H, W, m=2):
    """out (B, Cout, 2H, 2W) = A^T M A + bias from M ((m+2)^2, Cout, tiles); H, W low-res size."""
    Cout = M.shape[1]
    _chk("wino_output_transform", M, ((m + 2) ** 2, Cout, B * (2 * H // m) * (2 * W // m)))
    _chk("wino_output_transform", bias, (Cout,) if bias is not None else None)
    with torch.cuda.device(M.device):
        out = torch.empty((B, Cout, 2 * H, 2 * W), device=M.device, dtype=torch.float32)
        _lib.call("dmn_wino_output_transform", _ptr(M), _ptr(bias), _ptr(out), B, Cout, H, W, int(m),
                  _stream())
    return out


def head_final(x, feat, weff, bias0=None):
    """(B, 1, 2H, 2W) = conv3x3(upsample2x_bilinear(cat([x, feat], 1)), weff (C, 3, 3)) + bias0:
    the last head level composed with out_layer (see dmn_head_final).  Inference only."""
    B, Cx, H, W = x.shape
    Cf = feat.shape[1]
    _chk("head_final", x)
    _chk("head_final", feat, (B, Cf, H, W))
    _chk("head_final", weff, (Cx + Cf, 3, 3))
    _chk("head_final", bias0, (1,) if bias0 is not None else None)
    with torch.cuda.device(x.device):
        out = torch.empty((B, 1, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        _lib.call("dmn_head_final", _ptr(x), _ptr(feat), _ptr(weff), _ptr(bias0), _ptr(out),
                  B, Cx, Cf, H, W, _stream())
    return out


# ------------------------------------------------------------------------------------
# Head level as "tap GEMM at low resolution + gather at high resolution" (inference only)
# ------------------------------------------------------------------------------------
def tap_weight(weight):
    """Wt (9*Cout, C) with row (ky*3+kx)*Cout + o = w[o, :, ky, kx] of a (Cout, C, 3, 3) weight."""
    Cout, C = weight.shape[:2]
    return weight.permute(2, 3, 0, 1).reshape(9 * Cout, C).contiguous()


def tap_gemm(x, feat, wt_x, wt_f):
    """Z (B, 9, Cout, H, W) = Wt . cat([x, feat], 1) per sample, as one library GEMM per source
    (the second accumulates into the first's output, so the concat is never built)."""
    B, Cx, H, W = x.shape
    Cf = feat.shape[1]
    R = wt_x.shape[0]
    _chk("tap_gemm", x)
    _chk("tap_gemm", feat, (B, Cf, H, W))
    _chk("tap_gemm", wt_x, (R, Cx))
    _chk("tap_gemm", wt_f, (R, Cf))
    Z = None
    if Cx:
        Z = tuned_matmul(wt_x, x.view(B, Cx, H * W))
    if Cf:
        Z = tuned_matmul(wt_f, feat.view(B, Cf, H * W), acc=Z)
    return Z.view(B, 9, R // 9, H, W)


def tap_upsample_gather(Z, bias, out=None):
    """out (B, Cout, 2H, 2W) = bias + sum over the 9 taps of shift_tap(upsample2x(Z[:, tap]))
    (zero outside the high-res image); Z (B, 9, Cout, H, W).  `out` may be given: e.g. the first
    Cout channels of the next level's [x | feat] buffer (batch-strided view)."""
    B, nine, Cout, H, W = Z.shape
    _chk("tap_upsample_gather", Z, (B, 9, Cout, H, W))
    _chk("tap_upsample_gather", bias, (Cout,) if bias is not None else None)
    with torch.cuda.device(Z.device):
        if out is None:
            out = torch.empty((B, Cout, 2 * H, 2 * W), device=Z.device, dtype=torch.float32)
        bs = _chk_batch_strided("tap_upsample_gather", out, (B, Cout, 2 * H, 2 * W))
        _lib.call("dmn_tap_upsample_gather", _ptr(Z), _ptr(bias), _ptr(out), B, Cout, H, W, bs, _stream())
    return out


def tap_project(x, feat, w9):
    """Z (B, 9, 1, H, W) = w9 (9, Cx+Cf) . cat([x, feat], 1): the tap projection to ONE channel."""
    B, Cx, H, W = x.shape
    Cf = feat.shape[1]
    _chk("tap_project", x)
    _chk("tap_project", feat, (B, Cf, H, W))
    _chk("tap_project", w9, (9, Cx + Cf))
    with torch.cuda.device(x.device):
        Z = torch.empty((B, 9, 1, H, W), device=x.device, dtype=torch.float32)
        _lib.call("dmn_tap_project", _ptr(x) if Cx else None, _ptr(feat) if Cf else None, _ptr(w9),
                  _ptr(Z), B, Cx, Cf, H * W, _stream())
    return Z


# ------------------------------------------------------------------------------------
# DPN backbone glue (inference only: no autograd)
# ------------------------------------------------------------------------------------
def _chk_bn(name, consts, C):
    for t in consts:
        _chk(name, t, (C,))


def bn_relu(x, mean, scale, bias):
    """relu((x - mean[c]) * scale[c] + bias[c]) for x (N, C, H, W); eval-mode BN + ReLU."""
    N, C = x.shape[0], x.shape[1]
    HW = x.shape[2] * x.shape[3]
    _chk("bn_relu", x)
    _chk_bn("bn_relu", (mean, scale, bias), C)
    with torch.cuda.device(x.device):
        y = torch.empty_like(x)
        _lib.call("dmn_bn_relu", _ptr(x), _ptr(mean), _ptr(scale), _ptr(bias), _ptr(y),
                  N, C, HW, _stream())
    return y


def bn_relu_s2(x, mean, scale, bias):
    """relu(bn(x))[:, :, ::2, ::2] for x (N, C, H, W) with even H and W % 4 == 0."""
    N, C, H, W = x.shape
    bs = _chk_batch_strided("bn_relu_s2", x)                  # x may be a channel slice of a larger buffer
    _chk_bn("bn_relu_s2", (mean, scale, bias), C)
    with torch.cuda.device(x.device):
        y = torch.empty((N, C, H // 2, W // 2), device=x.device, dtype=torch.float32)
        _lib.call("dmn_bn_relu_s2", _ptr(x), _ptr(mean), _ptr(scale), _ptr(bias), _ptr(y), N, C, H, W,
                  bs, _stream())
    return y


def bn_relu_maxpool(x, mean, scale, bias):
    """MaxPool2d(3, 2, 1)(relu(bn(x))) for x (N, C, H, W) with W % 4 == 0: the stem's bn + act + pool."""
    N, C, H, W = x.shape
    _chk("bn_relu_maxpool", x)
    _chk_bn("bn_relu_maxpool", (mean, scale, bias), C)
    with torch.cuda.device(x.device):
        y = torch.empty((N, C, (H - 1) // 2 + 1, W // 2), device=x.device, dtype=torch.float32)
        _lib.call("dmn_bn_relu_maxpool", _ptr(x), _ptr(mean), _ptr(scale), _ptr(bias), _ptr(y), N, C, H, W,
                  _stream())
    return y


def channel_dot(x, w, bias0=None):
    """(N, 1, H, W) = sum_c w[c] * x[:, c] + bias0: a 1x1 convolution to one channel.  Inference."""
    N, C, H, W = x.shape
    _chk("channel_dot", x)
    _chk("channel_dot", w, (C,))
    _chk("channel_dot", bias0, (1,) if bias0 is not None else None)
    with torch.cuda.device(x.device):
        out = torch.empty((N, 1, H, W), device=x.device, dtype=torch.float32)
        _lib.call("dmn_channel_dot", _ptr(x), _ptr(w), _ptr(bias0), _ptr(out), N, C, H * W, _stream())
    return out


GCONV_GROUP_WIDTHS = (3, 4, 6, 8, 12, 16, 24, 32)


def gconv_weight_layout(weight, groups):
    """(C, C/groups, 3, 3) conv weight -> (groups, ci, 9, co): the layout dmn_gconv3x3_bn_relu
    reads (the co of one (ci, tap) adjacent, so they arrive as one scalar load)."""
    C, cg = weight.shape[:2]
    return weight.detach().reshape(groups, cg, cg, 9).permute(0, 2, 3, 1).contiguous()


def gconv3x3_bn_relu(x, weight_t, groups, bn_in=None, bn_out=None, stride=1):
    """relu(bn_out(conv3x3_grouped(relu(bn_in(x))))) in one pass; stride 1 or 2, pad 1, no bias.
    weight_t: gconv_weight_layout(conv.weight, groups); bn_in / bn_out: (mean, scale, bias) or
    None to skip that stage."""
    N, C, H, W = x.shape
    _chk("gconv3x3_bn_relu", x)
    weight = weight_t
    _chk("gconv3x3_bn_relu", weight, (groups, C // groups, 9, C // groups))
    for bn in (bn_in, bn_out):
        if bn is not None:
            _chk_bn("gconv3x3_bn_relu", bn, C)
    a = bn_in if bn_in is not None else (None, None, None)
    b = bn_out if bn_out is not None else (None, None, None)
    with torch.cuda.device(x.device):
        y = torch.empty((N, C, H // stride, W // stride), device=x.device, dtype=torch.float32)
        _lib.call("dmn_gconv3x3_bn_relu", _ptr(x), _ptr(weight), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]),
                  _ptr(b[0]), _ptr(b[1]), _ptr(b[2]), _ptr(y), N, C, H, W, int(groups), int(stride),
                  _stream())
    return y


def dpn_stage_pass(buf, c1c, inc, Co, bn, want_out=False, out=None):
    """In-place DPN stage buffer buf (N, rows, H, W) = [dense0 | res | slot | o2_1 ..] after the
    beta = 1 GEMM onto [res | slot]: returns (act, out) with act (N, Co, H, W) = relu(bn(X)) in the
    reference channel order and out = X itself when want_out (else None); moves the slot's
    channels to their final rows and zeroes the slot (see dmn_dpn_stage_pass).  `out` may be given
    (a batch-strided (N, Co, H, W) view, e.g. the feat channels of the head's [x | feat] buffer)."""
    N, rows, H, W = buf.shape
    _chk("dpn_stage_pass", buf)
    _chk_bn("dpn_stage_pass", bn, Co)
    with torch.cuda.device(buf.device):
        act = torch.empty((N, Co, H, W), device=buf.device, dtype=torch.float32)
        if out is None and want_out:
            out = torch.empty_like(act)
        bs = _chk_batch_strided("dpn_stage_pass", out, (N, Co, H, W)) if out is not None else Co * H * W
        _lib.call("dmn_dpn_stage_pass", _ptr(buf), _ptr(act), _ptr(out), _ptr(bn[0]), _ptr(bn[1]),
                  _ptr(bn[2]), N, rows, int(c1c), int(inc), int(Co), H * W, bs, _stream())
    return act, out


def dpn_merge(base, y, c1c, bn=None):
    """out = [base[:, :c1c] + y[:, :c1c] | base[:, c1c:] | y[:, c1c:]] in one pass.

    bn = (mean, scale, bias) of the next consumer's BatchNorm: also returns relu(bn(out)),
    written in the same pass; returns (out, act) (act None when bn is None).
    """
    N, Cb, H, W = base.shape
    Cy = y.shape[1]
    Co = Cb + Cy - c1c
    _chk("dpn_merge", base)
    _chk("dpn_merge", y, (N, Cy, H, W))
    if bn is not None:
        _chk_bn("dpn_merge", bn, Co)
    with torch.cuda.device(base.device):
        out = torch.empty((N, Co, H, W), device=base.device, dtype=torch.float32)
        act = torch.empty_like(out) if bn is not None else None
        mean, scale, bias = bn if bn is not None else (None, None, None)
        _lib.call("dmn_dpn_merge", _ptr(base), _ptr(y), _ptr(out), _ptr(act), _ptr(mean),
                  _ptr(scale), _ptr(bias), N, Cb, Cy, int(c1c), H * W, _stream())
    return out, act
