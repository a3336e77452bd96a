# -*- coding: utf-8 -*-
"""ctypes binding of libdmn_hip.so (the C ABI declared in include/dmn_hip.h).

There is deliberately NO fallback: if the shared library is missing or does not export a
symbol, importing the ops fails loudly -- a silently slower (or CPU) path would void every
parity and performance claim made for the HIP kernels.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMN_HIP_LIB", os.path.join(_HERE, "lib", "libdmn_hip.so"))
ABI_VERSION = 3

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_ll = ctypes.c_longlong

# name -> argtypes, mirrors include/dmn_hip.h one to one
SIGNATURES = {
    "dmn_sru_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p],
    "dmn_sru_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "dmn_sru_fwd_scan": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p],
    "dmn_sru_fwd_image": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _ll, _i, _p],
    "dmn_sru_fwd_scan_noalloc": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p],
    "dmn_sru_state_fwd": [_p, _p, _p, _p, _i, _i, _i, _p],
    "dmn_sru_state_fwd_train": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "dmn_sru_state_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "dmn_sru_state_fwd_scan": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "dmn_sru_scan_workspace_floats": [_i, _i, _i, _i],
    "dmn_bias_sigmoid": [_p, _p, _ll, _i, _p],
    "dmn_cat_coords": [_p, _p, _i, _i, _i, _i, _p],
    "dmn_dynfilter_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dmn_dynfilter_fwd_ws": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dmn_dynfilter_workspace_floats": [_i, _i, _i, _i, _i],
    "dmn_dynfilter_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dmn_mix_bcast_add": [_p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    "dmn_mix_sru_fwd": [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "dmn_mix_bcast_add_bwd": [_p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dmn_mix_bwd_parts": [_i, _i, _i, _i, _i],
    "dmn_mix_bwd_workspace_floats": [_i, _i, _i, _i, _i],
    "dmn_cat_upsample2x": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dmn_cat_upsample2x_bwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dmn_wino_cat_upsample2x_transform": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "dmn_wino_output_transform": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dmn_head_final": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dmn_tap_upsample_gather": [_p, _p, _p, _i, _i, _i, _i, _ll, _p],
    "dmn_tap_project": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "dmn_threshold_iou": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "dmn_gconv3x3_bn_relu": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "dmn_bn_relu": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "dmn_bn_relu_s2": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _ll, _p],
    "dmn_bn_relu_maxpool": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "dmn_channel_dot": [_p, _p, _p, _p, _i, _i, _i, _p],
    "dmn_dpn_stage_pass": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _ll, _p],
    "dmn_dpn_merge": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
}

# entry points whose return VALUE is not a C int
RESTYPES = {"dmn_mix_bwd_workspace_floats": _ll}

_lib = None

# Optional launch observer (bench.py installs one to bracket every entry-point call with
# HIP events on the launch stream); None in normal use.  begin(name, args) -> token, end(token).
TIMER = None


def load():
    """Load (once) and type the library; raises if it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "dms_amd: %s not found. Build it with `python -m dms_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.dmn_abi_version.restype = _i
    lib.dmn_last_error.restype = ctypes.c_char_p
    got = lib.dmn_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError("dms_amd: libdmn_hip.so has ABI %d, python side expects %d; rebuild"
                           % (got, ABI_VERSION))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: loud
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, _i)
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an entry point; non-zero status -> RuntimeError(dmn_last_error())."""
    lib = load()
    timer = TIMER
    token = timer.begin(name, args) if timer is not None else None
    rc = getattr(lib, name)(*args)
    if timer is not None:
        timer.end(token)
    if rc != 0:
        msg = lib.dmn_last_error()
        raise RuntimeError("%s failed (status %d): %s"
                           % (name, rc, msg.decode("utf-8", "replace") if msg else "?"))
    return rc


def query(name, *args):
    """Invoke an entry point that returns a VALUE (a size), not a status."""
    return int(getattr(load(), name)(*args))
