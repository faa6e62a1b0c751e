"""ctypes binding of libstereo_hip.so (include/stereo_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream;
every op is a direct C-ABI call with raw device pointers.  There is NO CPU or
eager fallback: if the library is missing or a tensor is not on the GPU the
call raises.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STEREO_HIP_LIB") or os.path.join(_HERE, "libstereo_hip.so")  # env: dev A/B builds only

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_ll = ctypes.c_longlong

# name -> argtypes (return type is int unless listed in _RESTYPE)
SIGNATURES = {
    "stereo_abi_version": [],
    "stereo_strerror": [_i],
    "stereo_test_option": [ctypes.c_char_p, _i],
    "stereo_corr_volume_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_diff_volume_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_concat_volume_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_gwc_volume_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_soft_argmin_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_disparity_regression_f32": [_vp, _vp, _i, _i, _i, _i, _vp],
    "stereo_group_mean_soft_argmin_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_corr_soft_argmin_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_corr_volume_soft_argmin_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_gwc_volume_soft_argmin_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_sample_volume_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_local_lookup_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "stereo_variance_lookup_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "stereo_bounds_lookup_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_uncertainty_volume_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "stereo_allpairs_pyramid_f32": [_vp, _vp, ctypes.POINTER(_vp), _i, _i, _i, _i, _i, _i, _vp],
    "stereo_corr_lookup_f32": [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_avg_pool_nhwc_f32": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_corr_lookup_nhwc_f32": [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_conv_small_cin_nhwc_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_alt_corr_lookup_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_alt_corr_workspace_floats": [_i, _i, _i, _i, _i],
    "stereo_bilinear_sampler_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_disp_warp_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_upsample_convex_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_upsample_convex_nhwc_mask_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_resize_bilinear_f32": [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "stereo_resize_bilinear_nhwc_f32": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "stereo_cross_attention_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_cross_attention_f16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_corr_volume_f16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_gwc_volume_f16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_corr_soft_argmin_f16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_corr_volume_soft_argmin_f16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_gwc_volume_soft_argmin_f16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_allpairs_pyramid_f16": [_vp, _vp, ctypes.POINTER(_vp), _i, _i, _i, _i, _i, _i, _vp],
    "stereo_instance_norm_workspace_bytes": [_i, _i],
    "stereo_instance_norm_relu_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "stereo_instance_norm_relu_nhwc": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "stereo_relayout_nhwc_to_nchw": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_relayout_nchw_to_nhwc_f32": [_vp, _vp, _i, _i, _i, _i, _vp],
    "stereo_hbm_arena_granularity": [ctypes.POINTER(_ll)],
    "stereo_hbm_arena_create": [_ll, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp)],
    "stereo_hbm_arena_keep": [_vp, _i, _i],
    "stereo_hbm_arena_destroy": [_vp],
    "stereo_conv3x3_small_nhwc_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_conv3x3_small_nhwc_f16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_conv3x3_small_nhwc_ex": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_conv_packed_bytes": [_i, _i, _i],
    "stereo_conv_packed_bytes_f16": [_i, _i, _i],
    "stereo_conv_packed_bytes_bf16x3": [_i, _i, _i],
    "stereo_conv_pack_weights_bf16x3": [_vp, _vp, _i, _i, _i, _vp],
    "stereo_conv_nhwc_f32_bf16x3": [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_i),
                                    ctypes.POINTER(_i), _i, _i, _i, _i, _i, _i, _vp],
    "stereo_conv_pack_weights": [_vp, _vp, _i, _i, _i, _vp],
    "stereo_conv_pack_weights_f16": [_vp, _vp, _i, _i, _i, _vp],
    "stereo_conv_nhwc_f16": [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_i),
                             ctypes.POINTER(_i), _i, _i, _i, _i, _i, _i, _vp],
    "stereo_conv_nhwc_f32": [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_i),
                             ctypes.POINTER(_i), _i, _i, _i, _i, _i, _i, _vp],
    "stereo_conv_packed_bytes_ex": [_i, _i, _i, _i],
    "stereo_conv_pack_weights_ex": [_i, _vp, _vp, _i, _i, _i, _vp],
    "stereo_conv_nhwc_ex": [_i, ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_i),
                            ctypes.POINTER(_i), _i, ctypes.POINTER(_i), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                            ctypes.POINTER(_vp), _i, _i, _i, _i, _i, _vp],
    "stereo_guidance_inputs_nhwc_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_fingerprint": [ctypes.POINTER(_vp), ctypes.POINTER(_ll), _i, _vp, _vp],
    "stereo_soft_argmin_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_soft_argmin_bwd_add_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_gwc_volume_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_local_lookup_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "stereo_local_lookup_bwd_dense_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "stereo_variance_lookup_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "stereo_variance_lookup_bwd_dense_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "stereo_diff_volume_bwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_concat_volume_bwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_upsample_convex_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_gru_reset_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp],
    "stereo_gru_blend_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp],
    "stereo_uncertainty_volume_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "stereo_disp_warp_bwd_disp_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stereo_bounds_lookup_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_corr_lookup_bwd_f32": [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_allpairs_pyramid_bwd_f32": [_vp, _vp, ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stereo_resize_bilinear_bwd_f32": [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "stereo_gru_reset_f32": [_vp, _vp, _vp, _vp, _ll, _vp],
    "stereo_gru_blend_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp],
    "stereo_gru_reset_f16": [_vp, _vp, _vp, _vp, _ll, _vp],
    "stereo_gru_blend_f16": [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp],
}
_RESTYPE = {"stereo_strerror": ctypes.c_char_p, "stereo_alt_corr_workspace_floats": _ll,
            "stereo_instance_norm_workspace_bytes": _ll, "stereo_conv_packed_bytes": _ll,
            "stereo_conv_packed_bytes_f16": _ll, "stereo_conv_packed_bytes_bf16x3": _ll,
            "stereo_conv_packed_bytes_ex": _ll}

_lib = None


class StereoHipError(RuntimeError):
    """Raised when a libstereo_hip.so entry point returns a STEREO_ERR_* code (``.code``)."""

    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


ERR_UNSUPPORTED = -3


def load() -> ctypes.CDLL:
    """dlopen the in-tree library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). stereoformer_amd has no CPU/eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    missing = [name for name in SIGNATURES if not hasattr(lib, name)]
    if missing:  # header/library mismatch: fail loudly, never degrade
        raise RuntimeError(f"{LIB_PATH} lacks symbols {missing}; rebuild it (stereoformer_amd/build.py)")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, _i)
    _lib = lib
    return lib


def check(rc: int, name: str) -> None:
    if rc != 0:
        msg = load().stereo_strerror(rc).decode()
        raise StereoHipError(f"{name} failed: {msg} (code {rc})", rc)


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)


@contextlib.contextmanager
def test_options(**opts):
    """TEST HOOK (``stereo_test_option``): steer which kernel serves the calls inside the ``with`` block, e.g.
    ``test_options(corr_mfma=0)`` runs the VALU fallback on a shape the matrix-core kernel would take.  Defaults are
    restored on exit.  Not part of the product surface; the library reads no environment variable."""
    lib = load()
    try:
        for k, v in opts.items():
            check(lib.stereo_test_option(k.encode(), int(v)), f"stereo_test_option({k})")
        yield
    finally:
        for k in opts:
            lib.stereo_test_option(k.encode(), -1)


def stream_ptr(device: torch.device | None = None) -> ctypes.c_void_p:
    return _vp(torch.cuda.current_stream(device).cuda_stream)


def ptr(t: torch.Tensor | None) -> ctypes.c_void_p:
    return _vp(0) if t is None else _vp(t.data_ptr())


def nchw(t: torch.Tensor) -> torch.Tensor:
    """NCHW-contiguous version of a HIP tensor.  A dense ``torch.channels_last`` fp32 / fp16 activation (what an extractor
    running in MIOpen's preferred layout hands over) goes through the library's tiled relayout kernel
    (``stereo_relayout_nhwc_to_nchw``: ~5 TB/s, where ``Tensor.contiguous()`` manages ~0.4 TB/s on this part); anything else
    takes ``contiguous()``."""
    if t.is_contiguous():
        return t
    if (t.dim() == 4 and t.is_cuda and t.dtype in (torch.float32, torch.float16)
            and t.is_contiguous(memory_format=torch.channels_last) and not (torch.is_grad_enabled() and t.requires_grad)):
        B, C, H, W = t.shape
        if C % 4 == 0 and C <= 256 and (H * W) % 4 == 0:
            out = torch.empty((B, C, H, W), device=t.device, dtype=t.dtype)
            h = 1 if t.dtype == torch.float16 else 0
            with torch.cuda.device(t.device):
                try:
                    call("stereo_relayout_nhwc_to_nchw", ptr(t), h, ptr(out), h, B, C, H, W, stream_ptr(t.device))
                    return out
                except StereoHipError as e:
                    if e.code != ERR_UNSUPPORTED:
                        raise
    return t.contiguous()


def nhwc(t: torch.Tensor) -> torch.Tensor:
    """``torch.channels_last`` version of a 4-D HIP tensor.  An NCHW-contiguous fp32 activation goes through the library's
    tiled relayout kernel (``stereo_relayout_nchw_to_nhwc_f32``; ``Tensor.contiguous(memory_format=...)`` moves ~1.3 TB/s on
    this part); anything else takes ``contiguous(memory_format=torch.channels_last)``."""
    cl = torch.channels_last
    if t.dim() != 4 or t.is_contiguous(memory_format=cl):
        return t if t.dim() != 4 else t.contiguous(memory_format=cl)
    if (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and not (torch.is_grad_enabled() and t.requires_grad)):
        B, C, H, W = t.shape
        if C % 32 == 0 and (H * W) % 4 == 0:
            out = torch.empty((B, C, H, W), device=t.device, dtype=t.dtype, memory_format=cl)
            with torch.cuda.device(t.device):
                try:
                    call("stereo_relayout_nchw_to_nhwc_f32", ptr(t), ptr(out), B, C, H, W, stream_ptr(t.device))
                    return out
                except StereoHipError as e:
                    if e.code != ERR_UNSUPPORTED:
                        raise
    return t.contiguous(memory_format=cl)


def dev_f32(t: torch.Tensor, name: str, dense_ok: bool = False) -> torch.Tensor:
    """Validate an op input: HIP device, fp32; returns it contiguous (NCHW).  ``dense_ok``: a channels-last dense
    tensor is returned as is -- for pointwise ops, which only need all operands in ONE dense layout."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name}: tensor is on {t.device}; stereoformer_amd runs only on an MI355X HIP device "
            "(no CPU fallback - the CPU restatement lives in oracle/ and is test-only)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype} (the path computes in f32)")
    if torch.is_grad_enabled() and t.requires_grad:
        raise RuntimeError(
            f"{name}: this op is forward-only (PytorchAlternateCorrBlock1D, build_cost_volume_from_volume, the fused "
            "inference entry points); every module surface of the path has a backward - call this one under torch.no_grad()")
    if dense_ok and t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t
    return nchw(t)


def dev_feat(t: torch.Tensor, name: str) -> torch.Tensor:
    """Validate a FEATURE operand of a matrix-core op: HIP device, fp32 or fp16 (an fp16 extractor hands
    its features over as they are; the kernels widen them on the way into LDS), contiguous NCHW."""
    if isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float16 and not (torch.is_grad_enabled() and t.requires_grad):
        return nchw(t)
    if isinstance(t, torch.Tensor) and t.dtype == torch.float16:
        t = t.float()          # grad mode: the differentiable ops are fp32 (autograd carries the cast)
    return dev_f32(t, name)


def feat_call(base: str, feats, args_of):
    """Call ``<base>_f16`` when every feature operand is fp16, else ``<base>_f32``.  The fp16 entry points cover the
    matrix-core envelope only: on STEREO_ERR_UNSUPPORTED the features are widened HERE (a device cast, not a fallback off
    the HIP path) and the fp32 entry point runs.  ``args_of(feats)`` builds the C argument tuple."""
    if all(f is None or f.dtype == torch.float16 for f in feats):
        try:
            return call(base + "_f16", *args_of(feats))
        except StereoHipError as e:
            if e.code != ERR_UNSUPPORTED:
                raise
    feats = [None if f is None else (f if f.dtype == torch.float32 else f.float()) for f in feats]
    return call(base + "_f32", *args_of(feats))


def one_layout(*ts):
    """Pointwise ops index the storage flat: every (non-None) operand must have the same strides.  Returns the
    operands unchanged when they do, NCHW-contiguous copies otherwise."""
    live = [t for t in ts if t is not None]
    if all(t.stride() == live[0].stride() for t in live):
        return ts
    return tuple(None if t is None else t.contiguous() for t in ts)


def same_device(*ts: torch.Tensor) -> torch.device:
    dev = ts[0].device
    for t in ts[1:]:
        if t is not None and t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev
