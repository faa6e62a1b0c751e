"""Stride-1 "same" convolutions of the update block (A14, ``update/update.py:6-33,66-91,130-178``) on the HIP path.

``conv_nhwc(srcs, convs, relu)`` is ``[m(torch.cat(srcs, 1)) for m in convs]`` (each ``m`` an ``nn.Conv2d`` with a 3x3 or 1x1
kernel, stride 1, padding k // 2, the same input) for channels-last fp32 activations, as ONE launch of
``csrc/conv_split.hip``: the concatenation is never materialised (the kernel's K loop walks the sources), the bias and an
optional ReLU ride in the epilogue, and several modules that read the same input (``convz`` | ``convr``; ``flow_head.conv1`` |
``mask[0]``) run as one convolution whose output channels are scattered to one tensor per module.  ``epilogue=`` puts the
ConvGRU's gate math (``update.py:28-32``) into the same launch: ``sigmoid(conv + c)``, ``sigmoid(conv + c) * h`` and
``(1 - z) * h + z * tanh(conv + c)`` instead of separate pointwise kernels over materialised pre-activations.

Arithmetic on fp32 activations (``precision``):

* ``"highest"`` (default, the parity configuration) -- every operand as two fp16 pieces ``x = hi + 2^-11 lo'`` (22-23 significant
  bits), the three leading piece products (each exact in fp32) on ``v_mfma_f32_32x32x16_f16``: fp32-faithful for
  ``|x| < 65504`` -- the range the reference's own ``mixed_precision`` flag assumes for this block; beyond it the result is
  inf / NaN, never silently wrong.  (= ``"f16x3"``: two fp32 accumulators, 4-row pixel tiles.)
* ``"f16x3s"`` -- the same three products into ONE accumulator: the weights are pre-scaled per output channel by a power of two
  and packed as three pieces; 8-row tiles with 64 channels per wave, half the LDS traffic per MFMA.  Same error budget; 7 %
  faster on the widest launch (384 -> 256 channels at [8,.,136,240]: 1.22 vs 1.31 ms), behind on narrow / small ones, equal
  over a whole RAFT loop (the kernel is power-bound either way) -- opt-in.
* ``"bf16x6"`` -- three bf16 pieces, six piece products: the same fidelity over fp32's whole exponent range at twice the matrix
  work (round 2's default).
* ``"high"`` -- two bf16 pieces, three products ("bf16x3", ~2^-16 relative error per product: the counterpart of
  ``torch.set_float32_matmul_precision('high')``).  NOT a parity configuration.

fp16 activations (the block under ``mixed_precision``) are the operands of the fp16 MFMA as they are.

Inference only (no autograd formula: the training path keeps the library convolutions); ``supported(...)`` tells a caller whether
its tensors are inside the kernel's envelope.

Packed weights are cached per module and keyed by what the host can see of a parameter (identity, storage address, version
counter, shape).  An in-place update through ``param.data`` (``m.weight.data.copy_(w)``, EMA swaps ...) changes none of those:
``refresh(root)`` compares CONTENT fingerprints computed on the device (one launch + one 8-byte-per-tensor read-back for a whole
module tree; the loops in ``pipelines.py`` call it once per forward) and drops what went stale; ``invalidate(root)`` drops
everything unconditionally."""
from __future__ import annotations

import ctypes
import threading

import torch

from . import _native as N

ARITH = {"bf16x6": 0, "bf16x3": 1, "f16": 2, "f16x3": 3, "f16x3s": 4}
_ALIAS = {"highest": "f16x3", "high": "bf16x3"}
EPILOGUES = {"linear": 0, "sigmoid": 1, "sigmoid_mul": 2, "gru_blend": 3, "gru_blend_softsign": 4}
_N_OPERANDS = {"linear": 0, "sigmoid": 1, "sigmoid_mul": 2, "gru_blend": 3, "gru_blend_softsign": 3}   # (addend, aux0, aux1) taken by each
_tls = threading.local()


def _form(precision):
    name = _ALIAS.get(precision, precision)
    if name not in ("f16x3", "f16x3s", "bf16x6", "bf16x3"):
        raise ValueError("conv_split: precision is 'highest' (= 'f16x3'), 'f16x3s', 'bf16x6' or 'high' (= 'bf16x3')")
    return name


def get_precision():
    return getattr(_tls, "precision", "highest")


def set_precision(precision):
    """Default arithmetic of ``conv_nhwc`` on fp32 activations FOR THE CALLING THREAD (see the module docstring); an explicit
    ``precision=`` argument wins.  Returns the previous setting."""
    _form(precision)
    prev = get_precision()
    _tls.precision = precision
    return prev


def _is_cl(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def supported(srcs, convs, pad_in=False) -> bool:
    """Channels-last fp32 (or all-fp16: the mixed_precision form) HIP tensors without grad, every source channel count a multiple of 16 (<= 4 sources), 3x3 / 1x1 stride-1
    'same' convolutions without groups or dilation, all but the last module with a multiple of 32 output channels, <= 640 in
    all.  ``pad_in``: the modules may take FEWER input channels than the sources carry (a source zero-padded to a multiple of 16:
    the missing weight columns are packed as zeros)."""
    if not (1 <= len(srcs) <= 4) or not convs:
        return False
    if torch.is_grad_enabled() and (any(t.requires_grad for t in srcs) or
                                    any(p.requires_grad for m in convs if isinstance(m, torch.nn.Module) for p in m.parameters())):
        return False     # no autograd formula: training keeps the library convolutions
    t0 = srcs[0]
    if not isinstance(t0, torch.Tensor) or t0.dtype not in (torch.float32, torch.float16):
        return False
    for t in srcs:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == t0.dtype and _is_cl(t)):
            return False
        if t.shape[0] != t0.shape[0] or t.shape[2:] != t0.shape[2:] or t.shape[1] % 16 != 0 or t.shape[1] < 16:
            return False
        if t.shape[1] * t.shape[2] * t.shape[3] * 4 >= 2 ** 31:
            return False
    cin = sum(t.shape[1] for t in srcs)
    k = convs[0].kernel_size
    total = 0
    for i, m in enumerate(convs):
        if not isinstance(m, torch.nn.Conv2d) or m.kernel_size != k or k not in ((3, 3), (1, 1)):
            return False
        if m.stride != (1, 1) or m.dilation != (1, 1) or m.groups != 1 or m.padding != (k[0] // 2, k[0] // 2):
            return False
        if (m.in_channels > cin or (m.in_channels != cin and not pad_in) or m.weight.dtype != torch.float32
                or m.padding_mode != "zeros"):
            return False
        if i + 1 < len(convs) and m.out_channels % 32 != 0:
            return False
        total += m.out_channels
    return total <= 640


# ---------------------------------------------------------------------------------------------- packed-weight cache
def _version(p):
    try:
        return p._version
    except RuntimeError:      # inference tensors carry no version counter
        return -1


def _params(mods):
    """The tensors a packed copy depends on: a convolution's weight / bias, a folded BatchNorm's affine parameters and running
    statistics."""
    out = []
    for m in mods:
        names = ("weight", "bias", "running_mean", "running_var") if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) \
            else ("weight", "bias")
        out += [getattr(m, n) for n in names if getattr(m, n, None) is not None]
    return out


def _signature(mods, device):
    """What the host can see of the parameters a packed copy depends on."""
    return (device.index,) + tuple((id(p), p.data_ptr(), _version(p), tuple(p.shape)) for p in _params(mods))


def fingerprints(tensors):
    """Content fingerprints (python ints) of fp32 HIP tensors: one launch (``stereo_fingerprint``) and one small read-back."""
    tensors = [t.detach() for t in tensors]
    if not tensors:
        return []
    dev = tensors[0].device
    for t in tensors:
        dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
        if not (t.is_cuda and t.device == dev and t.dtype == torch.float32 and dense):
            raise ValueError("fingerprints: dense fp32 tensors on one HIP device")
    out = torch.empty(len(tensors), device=dev, dtype=torch.int64)
    tp = (N._vp * len(tensors))(*[t.data_ptr() for t in tensors])
    nb = (N._ll * len(tensors))(*[t.numel() * 4 for t in tensors])
    with torch.cuda.device(dev):
        N.call("stereo_fingerprint", tp, nb, len(tensors), N.ptr(out), N.stream_ptr(dev))
    return [int(v) for v in out.tolist()]


def _cached(convs, tag, device, build):
    """Per-module cache (it lives on the first module of the group and dies with it -- a global table keyed by storage
    addresses would hand a freed weight's packed copy to whatever tensor is allocated at the same address next)."""
    store = convs[0].__dict__.setdefault("_stereo_conv_cache", {})
    key = (tag, device.index) + tuple(id(m) for m in convs[1:])   # (per device: nn.DataParallel replicas share this dict)
    sig = _signature(convs, device)
    hit = store.get(key)
    if hit is not None and hit[0] == sig:
        return hit[1]
    val = build()
    # [signature, value, the modules of the group, content fingerprints at build time (filled in by the first refresh() that
    # sees the entry BEFORE it is trusted across a forward -- see refresh)]
    store[key] = [sig, val, list(convs), fingerprints(_params(convs))
                  if not torch.cuda.is_current_stream_capturing() else None]
    return val


def invalidate(root):
    """Drop every packed copy cached on ``root`` (an ``nn.Module`` tree or an iterable of modules)."""
    mods = root.modules() if isinstance(root, torch.nn.Module) else root
    for m in mods:
        m.__dict__.pop("_stereo_conv_cache", None)


def refresh(root):
    """Re-validate the packed copies cached under ``root`` by CONTENT: fingerprints of all parameters involved are computed on
    the device in one launch and compared with those taken when each copy was built; stale copies are dropped (and rebuilt by
    the next call that needs them).  Catches in-place updates through ``param.data``, which bump no version counter.  Costs one
    small launch and one read-back per call -- made once per forward by the loops, outside any graph capture."""
    entries = []
    for m in (root.modules() if isinstance(root, torch.nn.Module) else root):
        store = m.__dict__.get("_stereo_conv_cache")
        if store:
            entries += [(store, k, e) for k, e in store.items()]
    if not entries or torch.cuda.is_current_stream_capturing():
        return 0
    by_dev = {}
    for store, k, e in entries:
        ps = _params(e[2])
        if ps and ps[0].is_cuda:
            by_dev.setdefault(ps[0].device, []).append((store, k, e, ps))
        else:
            store.pop(k, None)       # parameters moved off the device: the copy is dead
    dropped = 0
    for dev, items in by_dev.items():
        flat = [p for it in items for p in it[3]]
        fps = fingerprints(flat)
        pos = 0
        for store, k, e, ps in items:
            cur = fps[pos:pos + len(ps)]
            pos += len(ps)
            if e[3] is None or e[3] != cur:
                store.pop(k, None)
                dropped += 1
    return dropped


def _fold_bn(w, b, bn):
    """Inference-mode BatchNorm2d behind a convolution, folded into its weights: bn(conv(x)) = conv'(x) with
    w' = w * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(var + eps) + beta (fp64 arithmetic, rounded once)."""
    if bn.training or not bn.track_running_stats or bn.running_var is None:
        raise ValueError("conv_nhwc: only an eval-mode BatchNorm with running statistics can be folded into a convolution")
    s = torch.rsqrt(bn.running_var.detach().double() + bn.eps)
    if bn.weight is not None:
        s = s * bn.weight.detach().double()
    b0 = torch.zeros_like(s) if b is None else b.double()
    nb = (b0 - bn.running_mean.detach().double()) * s
    if bn.bias is not None:
        nb = nb + bn.bias.detach().double()
    return (w.double() * s.view(-1, 1, 1, 1)).float(), nb.float()


def _packed(convs, device, form, bns=None, cin=None, scales=None):
    """Packed weights of a group of modules in the operand form ``form`` (+ the concatenated fp32 bias); rebuilt when a parameter
    is replaced or updated in place.  ``bns``: per module an eval-mode BatchNorm2d folded into it (or None); ``cin``: input
    channels of the launch (>= the modules': the missing weight columns are zeros)."""
    arith = ARITH[form]
    bns = [None] * len(convs) if bns is None else list(bns)
    scales = [1.0] * len(convs) if scales is None else [float(x) for x in scales]

    def build():
        with torch.no_grad():
            ws, bs = [], []
            for m, bn, sc in zip(convs, bns, scales):
                w, b = m.weight.detach().contiguous(), None if m.bias is None else m.bias.detach()
                if bn is not None:
                    w, b = _fold_bn(w, b, bn)
                if sc != 1.0:    # (a constant factor on the module's output, e.g. the 0.25 of the mask head: folded, not launched)
                    w, b = w * sc, None if b is None else b * sc
                if cin is not None and w.shape[1] < cin:
                    w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cin - w.shape[1]))
                ws.append(w)
                bs.append(b)
            w = torch.cat(ws, 0).contiguous()
            bias = None
            if any(b is not None for b in bs):
                bias = torch.cat([b if b is not None else torch.zeros(m.out_channels, device=device)
                                  for b, m in zip(bs, convs)]).contiguous()
            cout, cin_, k = w.shape[0], w.shape[1], w.shape[2]
            nbytes = N.load().stereo_conv_packed_bytes_ex(arith, cout, cin_, k)
            if nbytes <= 0:
                raise ValueError(f"conv_nhwc: unsupported weight shape {tuple(w.shape)}")
            packed = torch.empty(nbytes, device=device, dtype=torch.uint8)
            with torch.cuda.device(device):
                N.call("stereo_conv_pack_weights_ex", arith, N.ptr(w), N.ptr(packed), cout, cin_, k, N.stream_ptr(device))
        return packed, bias, k
    return _cached(list(convs) + [bn for bn in bns if bn is not None], f"split_{form}_{cin}_{len(convs)}_{tuple(scales)}", device, build)


def conv_nhwc(srcs, convs, relu=False, out_channels_total=None, epilogue=None, precision=None, bns=None, pad_in=False,
              scales=None):
    """``[m(torch.cat(srcs, 1)) for m in convs]`` (optionally followed by ReLU), channels-last fp32 (or fp16: what the modules give
    under fp16 autocast) in and out, one launch.
    ``out_channels_total`` (single module only): allocate the output with that many channels and fill the first
    ``out_channels`` of them -- the caller owns the rest (the motion encoder appends the 2 flow channels, update.py:89-91).
    ``epilogue``: one entry per module, ``None`` (linear) or a tuple -- with ``t = m(cat(srcs)) + c`` (``c`` may be ``None``):
    ``("sigmoid", c)`` -> ``sigmoid(t)``; ``("sigmoid_mul", c, h)`` -> ``sigmoid(t) * h``; ``("gru_blend", c, h, z)`` ->
    ``(1 - z) * h + z * tanh(t)`` (``"gru_blend_softsign"``: ``t / (1 + |t|)`` for the tanh); ``("linear", c)`` -> ``t``.  The operand tensors have the output's shape, layout and dtype.
    ``precision``: see the module docstring (default: the calling thread's ``set_precision`` value).
    ``bns``: one eval-mode ``BatchNorm2d`` (or ``None``) per module, folded into the module's packed weights:
    ``bn(m(x))`` for the price of ``m(x)``.  ``pad_in``: the sources carry more channels than the modules take (a source
    zero-padded to a multiple of 16 channels); the extra channels meet zero weights.  ``scales``: a constant factor per module
    on its output (``0.25 * mask(x)``, update.py:176), folded into the packed weights and bias."""
    srcs = list(srcs)
    convs = list(convs)
    if bns is not None and (len(bns) != len(convs) or any(b is not None and b.num_features != m.out_channels for b, m in zip(bns, convs))):
        raise ValueError("conv_nhwc: one BatchNorm2d (or None) per module, over the module's output channels")
    if not supported(srcs, convs, pad_in=pad_in):
        raise ValueError("conv_nhwc: operands outside the kernel's envelope (see conv_split.supported)")
    if out_channels_total is not None and (len(convs) != 1 or out_channels_total < convs[0].out_channels):
        raise ValueError("conv_nhwc: out_channels_total applies to one module and must cover its output channels")
    dev = srcs[0].device
    half = srcs[0].dtype == torch.float16
    form = "f16" if half else _form(get_precision() if precision is None else precision)
    if scales is not None and len(scales) != len(convs):
        raise ValueError("conv_nhwc: one scale per module")
    packed, bias, k = _packed(convs, dev, form, bns, sum(t.shape[1] for t in srcs) if pad_in else None, scales)
    B, _, H, W = srcs[0].shape
    widths = [m.out_channels for m in convs] if out_channels_total is None else [int(out_channels_total)]
    outs = [torch.empty((B, c, H, W), device=dev, dtype=srcs[0].dtype, memory_format=torch.channels_last) for c in widths]
    n = len(outs)
    epi = (N._i * n)(*([0] * n))
    operands = [[None] * n for _ in range(3)]
    if epilogue is not None:
        if len(epilogue) != n:
            raise ValueError("conv_nhwc: one epilogue entry per module")
        for j, e in enumerate(epilogue):
            if e is None:
                continue
            kind, ops = e[0], list(e[1:])
            if kind not in EPILOGUES or len(ops) != _N_OPERANDS[kind] + (kind == "linear"):
                raise ValueError(f"conv_nhwc: bad epilogue {e[0]!r} / operand count")
            if kind != "linear" and relu:
                raise ValueError("conv_nhwc: relu applies to linear epilogues only")
            for i, t in enumerate(ops):
                if t is None:
                    if i > 0:
                        raise ValueError(f"conv_nhwc: epilogue {kind!r} needs its tensor operands")
                    continue
                if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == dev and t.dtype == outs[j].dtype
                        and t.shape == outs[j].shape and _is_cl(t)):
                    raise ValueError(f"conv_nhwc: epilogue operands must match the output ({tuple(outs[j].shape)}, "
                                     f"{outs[j].dtype}, channels-last)")
                operands[i][j] = t
            epi[j] = EPILOGUES[kind]
    sp = (N._vp * len(srcs))(*[t.data_ptr() for t in srcs])
    sc = (N._i * len(srcs))(*[t.shape[1] for t in srcs])
    op = (N._vp * n)(*[t.data_ptr() for t in outs])
    oc = (N._i * n)(*[m.out_channels for m in convs])
    os_ = (N._i * n)(*widths)
    aux = [(N._vp * n)(*[None if t is None else t.data_ptr() for t in row]) for row in operands]
    with torch.cuda.device(dev):
        N.call("stereo_conv_nhwc_ex", ARITH[form], sp, sc, len(srcs), N.ptr(packed), N.ptr(bias), op, oc, os_, n, epi, aux[0],
               aux[1], aux[2], k, 1 if relu else 0, B, H, W, N.stream_ptr(dev))
    return outs


def small_supported(x, conv) -> bool:
    """3x3 stride-1 'same' convolution with <= 4 output channels over a channels-last fp32 HIP tensor, Cin % 32 == 0."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in conv.parameters())):
        return False
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros" and conv.out_channels <= 4
            and conv.weight.dtype == torch.float32 and x.is_cuda and x.dtype in (torch.float32, torch.float16) and _is_cl(x)
            and x.shape[1] == conv.in_channels and x.shape[1] % 32 == 0 and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 2 ** 31)


def conv3x3_small(x, conv, relu=False):
    """``conv(x)`` (``relu(conv(x))``) for a 3x3 convolution with <= 4 output channels (``flow_head.conv2``, the offset head's
    ``conv2``): channels-last fp32 / fp16 in, NCHW-contiguous fp32 out."""
    if not small_supported(x, conv):
        raise ValueError("conv3x3_small: operands outside the kernel's envelope (see conv_split.small_supported)")
    half = x.dtype == torch.float16   # mixed_precision: the weights rounded to fp16 as autocast hands them to the library
    wt = _cached([conv], "tco16" if half else "tco", x.device,
                 lambda: (conv.weight.detach().half().float() if half else conv.weight.detach()).permute(2, 3, 1, 0).contiguous())
    B, C, H, W = x.shape
    out = torch.empty((B, conv.out_channels, H, W), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        N.call("stereo_conv3x3_small_nhwc_ex", N.ptr(x), 1 if half else 0, N.ptr(wt),
               N.ptr(None if conv.bias is None else conv.bias.detach()),
               N.ptr(out), B, C, conv.out_channels, H, W, 1 if relu else 0, N.stream_ptr(x.device))
    return out


def small_cin_supported(x, conv) -> bool:
    """KxK (3 / 5 / 7) stride-1 'same' convolution over <= 4 input channels of an NCHW fp32 HIP tensor, 16 / 32 / 48 / 64 output
    channels."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in conv.parameters())):
        return False
    k = conv.kernel_size[0] if isinstance(conv, torch.nn.Conv2d) else 0
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (k, k) and k in (3, 5, 7) and conv.stride == (1, 1)
            and conv.padding == (k // 2, k // 2) and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and conv.in_channels <= 4 and conv.out_channels % 16 == 0 and conv.out_channels <= 64
            and conv.weight.dtype == torch.float32 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[1] == conv.in_channels and x.is_contiguous())


def conv_small_cin(x, conv, relu=False):
    """``conv(x)`` (``relu(conv(x))``) for a KxK convolution over a handful of input channels (``convf1``: 7x7 over the 2-channel
    flow, update.py:79,87): NCHW fp32 in, channels-last fp32 out.  The 7x7 / 2-channel / 32- or 64-output case (``convf1`` itself)
    runs as an implicit GEMM on the fp16 matrix cores with split operands (fp32-faithful for ``|x| < 4094``: inf / NaN beyond,
    never silently wrong); everything else on the fp32 vector units."""
    if not small_cin_supported(x, conv):
        raise ValueError("conv_small_cin: operands outside the kernel's envelope (see conv_split.small_cin_supported)")
    wt = _cached([conv], "tco_small_cin", x.device, lambda: conv.weight.detach().permute(2, 3, 1, 0).contiguous())
    B, C, H, W = x.shape
    out = torch.empty((B, conv.out_channels, H, W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        N.call("stereo_conv_small_cin_nhwc_f32", N.ptr(x), N.ptr(wt), N.ptr(None if conv.bias is None else conv.bias.detach()),
               N.ptr(out), B, C, conv.out_channels, H, W, conv.kernel_size[0], conv.out_channels, 1 if relu else 0,
               N.stream_ptr(x.device))
    return out
