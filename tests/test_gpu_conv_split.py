"""GPU parity of the update block's convolutions on the split-operand matrix-core kernel (csrc/conv_split.hip) against the fp64
CPU convolution of the same operands (the oracle for a library op is the framework's own CPU kernel in double precision) and
against MIOpen's fp32 result on the device."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    import stereoformer_amd as s
    s._native.load()
    return s


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _ref64(srcs, convs, relu):
    x = torch.cat([s.double().cpu() for s in srcs], 1)
    outs = []
    for m in convs:
        y = F.conv2d(x, m.weight.detach().double().cpu(), None if m.bias is None else m.bias.detach().double().cpu(),
                     padding=m.padding)
        outs.append(F.relu(y) if relu else y)
    return outs


@pytest.mark.parametrize("case", [
    # (B, H, W, source channels, output channels per module, kernel, relu, bias)
    (1, 8, 32, (16,), (32,), 3, False, True),
    (2, 11, 37, (16, 32), (128, 40), 3, True, True),
    (1, 17, 60, (128, 128, 128), (128, 128), 3, False, True),       # convz | convr of a GRU
    (1, 9, 40, (64, 64), (126,), 3, True, True),                      # the motion encoder's last convolution
    (2, 5, 33, (128,), (256, 256), 3, True, False),                  # flow_head.conv1 | mask[0]
    (1, 6, 70, (32, 16, 16, 64), (24,), 1, False, True),              # 1x1
    (1, 34, 60, (128, 128), (128,), 3, False, True),
])
@pytest.mark.parametrize("precision", ["f16x3", "f16x3-rows4", "f16x3s", "bf16x6"])
def test_conv_nhwc_matches_fp64_and_miopen(S, case, precision):
    from stereoformer_amd import conv_split as C
    if precision == "f16x3-rows4":     # small launches of the default form take 2-row pixel tiles: force the 4-row shape too
        with S._native.test_options(conv_rows=4):
            return test_conv_nhwc_matches_fp64_and_miopen(S, case, "f16x3")
    B, H, W, cs, co, k, relu, bias = case
    torch.manual_seed(81)
    srcs = [_cl(torch.randn(B, c, H, W, device="cuda")) for c in cs]
    convs = [nn.Conv2d(sum(cs), c, k, padding=k // 2, bias=bias).cuda() for c in co]
    assert not C.supported(srcs, convs)          # grad mode with trainable weights: the library path
    with torch.no_grad():
        assert C.supported(srcs, convs)
        outs = C.conv_nhwc(srcs, convs, relu=relu, precision=precision)
        lib = [m(torch.cat(srcs, 1)) for m in convs]
    ref = _ref64(srcs, convs, relu)
    for o, r, l, m in zip(outs, ref, lib, convs):
        assert o.shape == r.shape and o.is_contiguous(memory_format=torch.channels_last)
        scale = float(r.abs().max())
        err = float((o.double().cpu() - r).abs().max())
        err_lib = float(((F.relu(l) if relu else l).double().cpu() - r).abs().max())
        # fp32-faithful: within a few ulps of the fp64 result at the output's scale (split-fp16: two fp32 accumulators, three
        # accumulations per 16 input channels; split-bf16: six into one -- the same order of rounding error as the library's)
        assert err <= 4e-6 * scale + 1e-6, (err, scale)
        assert err <= 10 * err_lib + 1e-6, (err, err_lib)


def test_conv_nhwc_repacks_after_an_in_place_weight_update_and_refuses_outside_its_envelope(S):
    from stereoformer_amd import conv_split as C
    torch.manual_seed(82)
    x = _cl(torch.randn(1, 16, 8, 32, device="cuda"))
    m = nn.Conv2d(16, 32, 3, padding=1).cuda()
    with torch.no_grad():
        a = C.conv_nhwc([x], [m])[0]
        m.weight.mul_(2.0)
        m.bias.zero_()
        b = C.conv_nhwc([x], [m])[0]
        assert torch.allclose(b, m(x), rtol=1e-5, atol=1e-5) and not torch.allclose(a, b)
        assert not C.supported([x.contiguous()], [m])                                   # NCHW
        assert not C.supported([x], [nn.Conv2d(16, 32, 3, padding=1, stride=2).cuda()])  # strided
        assert not C.supported([x[:, :8]], [nn.Conv2d(8, 32, 3, padding=1).cuda()])      # channels not a multiple of 16
        with pytest.raises(ValueError):
            C.conv_nhwc([x.detach().contiguous()], [m])
    assert not C.supported([x], [m])                                                    # training keeps the library path


def test_resize_bilinear_channels_last(S):
    torch.manual_seed(83)
    for (n, c, hi, wi, ho, wo, ac) in ((2, 128, 9, 15, 17, 30, True), (1, 8, 5, 7, 40, 56, True), (1, 4, 6, 6, 3, 9, False)):
        x = torch.randn(n, c, hi, wi, device="cuda")
        a = S.upsample.resize_bilinear(x, (ho, wo), ac, 1.5)
        b = S.upsample.resize_bilinear(_cl(x), (ho, wo), ac, 1.5)
        assert b.is_contiguous(memory_format=torch.channels_last)
        assert torch.allclose(a, b.contiguous(), rtol=1e-6, atol=1e-6)   # same formula; FMA contraction may differ by an ulp


def test_update_block_on_the_split_convolutions_matches_the_library_path(S, monkeypatch):
    """BasicMultiUpdateBlock_LZH in torch.channels_last under no_grad takes conv_split for its 3x3 convolutions (GRU gates as
    one convz|convr launch over the un-concatenated sources, motion encoder, flow / mask heads): same outputs as the library
    path (torch.cat + MIOpen) to fp32 rounding, over several chained iterations."""
    from stereoformer_amd import conv_split as C
    torch.manual_seed(84)
    B, H, W = 2, 24, 40
    blk = S.BasicMultiUpdateBlock_LZH(4, 4, 3, 2, [128, 128, 128]).cuda().eval().to(memory_format=torch.channels_last)
    mk = lambda c, h, w: _cl(torch.randn(B, c, h, w, device="cuda") * 0.5)
    nets0 = [torch.tanh(mk(128, H, W)), torch.tanh(mk(128, H // 2, W // 2)), torch.tanh(mk(128, H // 4, W // 4))]
    inps = [[mk(128, H, W) for _ in range(3)], [mk(128, H // 2, W // 2) for _ in range(3)], [mk(128, H // 4, W // 4) for _ in range(3)]]
    corr = torch.randn(B, 36, H, W, device="cuda")
    flow = torch.randn(B, 2, H, W, device="cuda")
    calls = []
    real = C.conv_nhwc
    monkeypatch.setattr(C, "conv_nhwc", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run(split):
        if not split:
            monkeypatch.setattr(C, "supported", lambda *a, **k: False)
        nets = [n.clone() for n in nets0]
        with torch.no_grad():
            for it in range(3):
                nets, mask, delta = blk(nets, inps, corr, flow + 0.1 * it, need_mask=it == 2)
        return nets, mask, delta

    a = run(True)
    n_split = len(calls)
    assert n_split == 3 * (3 * 2 + 3 + 1) + 1      # per iteration: 3 GRUs x 2 launches, 3 encoder convolutions, the heads; + mask[2]
    b = run(False)
    assert len(calls) == n_split
    for x, y in zip([*a[0], a[1], a[2]], [*b[0], b[1], b[2]]):
        assert x.shape == y.shape
        assert float((x - y).abs().max()) <= 2e-5 * max(1.0, float(y.abs().max())), float((x - y).abs().max())


@pytest.mark.parametrize("case", [(2, 32, 9, 37, 2), (1, 256, 17, 70, 2), (1, 64, 8, 32, 4), (1, 96, 5, 5, 1), (2, 128, 3, 100, 3)])
def test_conv3x3_small_output_channels(S, case):
    """flow_head.conv2 (256 -> 2) on the fp32 vector kernel: channels-last in, NCHW out, against fp64."""
    from stereoformer_amd import conv_split as C
    B, cin, H, W, cout = case
    torch.manual_seed(85)
    x = _cl(torch.randn(B, cin, H, W, device="cuda"))
    m = nn.Conv2d(cin, cout, 3, padding=1).cuda()
    with torch.no_grad():
        assert C.small_supported(x, m)
        o = C.conv3x3_small(x, m)
        assert not C.small_supported(x, nn.Conv2d(cin, 8, 3, padding=1).cuda())
    r = F.conv2d(x.double().cpu(), m.weight.detach().double().cpu(), m.bias.detach().double().cpu(), padding=1)
    assert o.shape == r.shape and o.is_contiguous()
    assert float((o.double().cpu() - r).abs().max()) <= 5e-6 * float(r.abs().max()) + 2e-6   # sequential fp32 sums of 9 Cin terms


@pytest.mark.parametrize("case", [(2, 11, 37, (16, 32), (128, 40), 3, True), (1, 17, 60, (128, 128, 128), (128, 128), 3, False),
                                  (1, 6, 70, (32, 16, 16, 64), (24,), 1, False)])
def test_conv_nhwc_fp16_activations(S, case):
    """The mixed_precision form: fp16 channels-last activations, weights rounded to fp16, fp32 accumulation, one rounding of
    the output -- against the fp64 convolution of the SAME rounded operands (error = the output's fp16 rounding) and against the
    library's fp16 convolution under autocast."""
    from stereoformer_amd import conv_split as C
    B, H, W, cs, co, k, relu = case
    torch.manual_seed(86)
    srcs = [_cl(torch.randn(B, c, H, W, device="cuda").half()) for c in cs]
    convs = [nn.Conv2d(sum(cs), c, k, padding=k // 2).cuda() for c in co]
    with torch.no_grad():
        assert C.supported(srcs, convs)
        outs = C.conv_nhwc(srcs, convs, relu=relu)
        with torch.autocast("cuda", dtype=torch.float16):
            lib = [m(torch.cat(srcs, 1)) for m in convs]
    x = torch.cat([s.double().cpu() for s in srcs], 1)
    for o, l, m in zip(outs, lib, convs):
        r = F.conv2d(x, m.weight.detach().half().double().cpu(), m.bias.detach().double().cpu(), padding=m.padding)
        r = F.relu(r) if relu else r
        assert o.dtype == torch.float16 and o.is_contiguous(memory_format=torch.channels_last)
        scale = float(r.abs().max())
        assert float((o.double().cpu() - r).abs().max()) <= 6e-4 * scale + 1e-4          # half an fp16 ulp at the output's scale
        assert float((o.float() - (F.relu(l) if relu else l).float()).abs().max()) <= 2e-3 * scale + 1e-3


def test_conv3x3_small_fp16_activations(S):
    from stereoformer_amd import conv_split as C
    torch.manual_seed(87)
    x = _cl(torch.randn(2, 64, 9, 37, device="cuda").half())
    m = nn.Conv2d(64, 2, 3, padding=1).cuda()
    with torch.no_grad():
        o = C.conv3x3_small(x, m)
    r = F.conv2d(x.double().cpu(), m.weight.detach().half().double().cpu(), m.bias.detach().double().cpu(), padding=1)
    assert o.dtype == torch.float32 and o.is_contiguous()
    assert float((o.double().cpu() - r).abs().max()) <= 5e-6 * float(r.abs().max()) + 2e-6


def test_conv_nhwc_high_precision_option(S):
    """conv_precision='high' (bf16x3: two pieces, three products): error ~2^-16 relative per product -- far inside TF32's, outside
    the default form's; opt-in and restored."""
    from stereoformer_amd import conv_split as C
    torch.manual_seed(88)
    srcs = [_cl(torch.randn(1, 128, 17, 60, device="cuda")) for _ in range(3)]
    convs = [nn.Conv2d(384, 128, 3, padding=1).cuda() for _ in range(2)]
    ref = _ref64(srcs, convs, False)
    with torch.no_grad():
        hi = C.conv_nhwc(srcs, convs)
        prev = C.set_precision("high")
        try:
            assert prev == "highest"
            lo = C.conv_nhwc(srcs, convs)
        finally:
            C.set_precision(prev)
        again = C.conv_nhwc(srcs, convs)
    for a, b, c, r in zip(hi, lo, again, ref):
        scale = float(r.abs().max())
        e_hi, e_lo = float((a.double().cpu() - r).abs().max()), float((b.double().cpu() - r).abs().max())
        assert e_hi <= 4e-6 * scale + 1e-6
        assert e_hi < e_lo <= 2e-4 * scale           # coarser than the default, ~30x finer than TF32 would be
        assert torch.equal(a, c)


def _gru_ref64(h, xs, cz, cr, cq, convz, convr, convq, softsign=False):
    """update.py:28-32 in fp64 on the CPU."""
    d = lambda t: t.double().cpu()
    conv = lambda m, x: F.conv2d(x, d(m.weight.detach()), d(m.bias.detach()), padding=m.padding)
    hx = torch.cat([d(h), *[d(x) for x in xs]], 1)
    z = torch.sigmoid(conv(convz, hx) + (0 if cz is None else d(cz)))
    r = torch.sigmoid(conv(convr, hx) + (0 if cr is None else d(cr)))
    t = conv(convq, torch.cat([r * d(h), *[d(x) for x in xs]], 1)) + (0 if cq is None else d(cq))
    q = t / (1 + t.abs()) if softsign else torch.tanh(t)
    return z, r * d(h), (1 - z) * d(h) + z * q


@pytest.mark.parametrize("case", [(2, 13, 40, 128, (128, 128), True, False), (1, 9, 33, 128, (128,), True, False),
                                  (1, 7, 20, 64, (32,), False, True), (2, 6, 18, 64, (32,), False, False)])
@pytest.mark.parametrize("precision", ["f16x3", "f16x3-rows4", "f16x3s", "bf16x6"])
def test_conv_nhwc_gru_epilogues(S, case, precision):
    if precision == "f16x3-rows4":
        with S._native.test_options(conv_rows=4):
            return test_conv_nhwc_gru_epilogues(S, case, "f16x3")
    """The ConvGRU as two launches with the gate math in the epilogues (sigmoid -> z, sigmoid x h -> r h, tanh / softsign + blend)
    against the fp64 statement of update.py:28-32, and bit-for-bit against the unfused form (linear epilogue + the pointwise
    gate kernels): both evaluate the same device functions on the same accumulator values."""
    from stereoformer_amd import conv_split as C
    from stereoformer_amd.update_dlc import gru_blend, gru_reset
    B, H, W, hd, xcs, ctx, softsign = case
    torch.manual_seed(91)
    h = _cl(torch.tanh(torch.randn(B, hd, H, W, device="cuda")))
    xs = [_cl(torch.randn(B, c, H, W, device="cuda")) for c in xcs]
    cz, cr, cq = [(_cl(torch.randn(B, hd, H, W, device="cuda")) if ctx else None) for _ in range(3)]
    cin = hd + sum(xcs)
    convz, convr, convq = [nn.Conv2d(cin, hd, 3, padding=1).cuda() for _ in range(3)]
    blend = "gru_blend_softsign" if softsign else "gru_blend"
    with torch.no_grad():
        if C.supported([h, *xs], [convz, convr]):
            z, gated = C.conv_nhwc([h, *xs], [convz, convr], epilogue=[("sigmoid", cz), ("sigmoid_mul", cr, h)], precision=precision)
        else:
            (z,) = C.conv_nhwc([h, *xs], [convz], epilogue=[("sigmoid", cz)], precision=precision)
            (gated,) = C.conv_nhwc([h, *xs], [convr], epilogue=[("sigmoid_mul", cr, h)], precision=precision)
        (hn,) = C.conv_nhwc([gated, *xs], [convq], epilogue=[(blend, cq, h, z)], precision=precision)
        # unfused: the same convolutions with linear epilogues, then the pointwise kernels
        (zp,), (rp,) = C.conv_nhwc([h, *xs], [convz], precision=precision), C.conv_nhwc([h, *xs], [convr], precision=precision)
        gated_u = gru_reset(h, rp, cr)
        (qp,) = C.conv_nhwc([gated_u, *xs], [convq], precision=precision)
        hn_u = gru_blend(h, zp, qp, cz, cq, softsign=softsign)
    assert torch.equal(gated, gated_u) and torch.equal(hn, hn_u)
    for t in (z, gated, hn):
        assert t.is_contiguous(memory_format=torch.channels_last) and t.shape == h.shape
    rz, rg, rh = _gru_ref64(h, xs, cz, cr, cq, convz, convr, convq, softsign)
    for got, ref in ((z, rz), (gated, rg), (hn, rh)):
        assert float((got.double().cpu() - ref).abs().max()) <= 3e-6, float((got.double().cpu() - ref).abs().max())


def test_conv_nhwc_epilogue_argument_checks(S):
    from stereoformer_amd import conv_split as C
    torch.manual_seed(92)
    x = _cl(torch.randn(1, 16, 8, 32, device="cuda"))
    m = nn.Conv2d(16, 32, 3, padding=1).cuda()
    h = _cl(torch.randn(1, 32, 8, 32, device="cuda"))
    with torch.no_grad():
        with pytest.raises(ValueError):
            C.conv_nhwc([x], [m], epilogue=[("sigmoid_mul", None, None)])          # missing operand
        with pytest.raises(ValueError):
            C.conv_nhwc([x], [m], epilogue=[("sigmoid", h.contiguous())])         # NCHW operand
        with pytest.raises(ValueError):
            C.conv_nhwc([x], [m], epilogue=[("sigmoid", h[:, :16])])              # wrong shape
        with pytest.raises(ValueError):
            C.conv_nhwc([x], [m], relu=True, epilogue=[("sigmoid", None)])        # ReLU with a gate epilogue
        with pytest.raises(ValueError):
            C.conv_nhwc([x], [m], epilogue=[("swish", None)])
        with pytest.raises(ValueError):
            C.conv_nhwc([x], [m], precision="fp8")
        (a,) = C.conv_nhwc([x], [m], epilogue=[("linear", h)])
        assert torch.allclose(a, m(x) + h, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("scale", [2.0 ** -12, 2.0 ** -6, 1.0, 2.0 ** 9])
def test_conv_nhwc_split_fp16_is_scale_invariant_inside_the_fp16_range(S, scale):
    """The default arithmetic splits every operand into two fp16 pieces with the low piece scaled by 2^11: its RELATIVE error
    must not depend on the magnitude of the data inside fp16's normal range, 2^-14 <= |x| < 65504 (activations x 2^-12 .. 2^9,
    weights x 2^-3.6 .. 2^2.7); below it the error is absolute, <= 2^-36 per operand."""
    from stereoformer_amd import conv_split as C
    torch.manual_seed(93)
    srcs = [_cl(torch.randn(1, 64, 12, 40, device="cuda") * scale) for _ in range(2)]
    m = nn.Conv2d(128, 64, 3, padding=1).cuda()
    with torch.no_grad():
        m.weight.mul_(scale ** 0.3)
        m.bias.zero_()
        (o,) = C.conv_nhwc(srcs, [m], precision="highest")
    (r,) = _ref64(srcs, [m], False)
    err, mag = float((o.double().cpu() - r).abs().max()), float(r.abs().max())
    assert err <= 4e-6 * mag, (err, mag)


def test_conv_nhwc_split_fp16_overflow_is_loud(S):
    """Beyond the fp16 range the default form's result is non-finite -- never silently wrong; 'bf16x6' covers fp32's range."""
    from stereoformer_amd import conv_split as C
    torch.manual_seed(94)
    x = _cl(torch.randn(1, 16, 8, 32, device="cuda"))
    x[0, 3, 4, 5] = 7.0e4                       # beyond fp16's largest finite value
    m = nn.Conv2d(16, 32, 3, padding=1).cuda()
    with torch.no_grad():
        (o,) = C.conv_nhwc([x], [m], precision="highest")
        (b,) = C.conv_nhwc([x], [m], precision="bf16x6")
        assert not bool(torch.isfinite(o[0, :, 3:6, 4:7]).any())          # every output that saw the value
        assert bool(torch.isfinite(o[0, :, :2]).all())
        assert torch.allclose(b, m(x), rtol=1e-5, atol=1e-2)             # the split-bf16 form covers fp32's range


def test_packed_weights_follow_updates_through_dot_data(S):
    """`param.data.copy_()` / `.data.mul_()` leave id, address, version counter and shape unchanged: refresh() finds the change
    by content (device fingerprints); RAFTStereoLoop / LowCNNGRULoop call it once per forward."""
    from stereoformer_amd import conv_split as C
    torch.manual_seed(95)
    x = _cl(torch.randn(1, 16, 8, 32, device="cuda"))
    m = nn.Conv2d(16, 32, 3, padding=1).cuda()
    with torch.no_grad():
        a = C.conv_nhwc([x], [m])[0]
        v = m.weight._version
        m.weight.data.mul_(2.0)
        assert m.weight._version == v                                   # invisible to the host-side key
        assert C.refresh([m]) == 1
        b = C.conv_nhwc([x], [m])[0]
        assert torch.allclose(b, m(x), rtol=1e-5, atol=1e-5) and not torch.allclose(a, b)
        assert C.refresh([m]) == 0                                      # nothing changed: nothing dropped
        c = C.conv_nhwc([x], [m])[0]
        assert torch.equal(b, c)
        m.bias.data.copy_(torch.randn_like(m.bias))
        assert C.refresh(nn.Sequential(m)) == 1
        assert torch.allclose(C.conv_nhwc([x], [m])[0], m(x), rtol=1e-5, atol=1e-5)
        C.invalidate([m])
        assert "_stereo_conv_cache" not in m.__dict__
        fa, fb = C.fingerprints([m.weight, m.bias]), C.fingerprints([m.weight.flip(0).contiguous(), m.bias])
        assert fa[1] == fb[1] and fa[0] != fb[0]                        # position-dependent


def test_loop_follows_dot_data_weight_updates_with_and_without_graph(S):
    from stereoformer_amd import conv_split as C
    torch.manual_seed(96)
    dev = torch.device("cuda")
    loop = S.RAFTStereoLoop([128, 128, 128]).eval().to(dev)
    B, Hq, Wq = 1, 16, 32
    f1, f2 = torch.randn(B, 256, Hq, Wq, device=dev), torch.randn(B, 256, Hq, Wq, device=dev)
    nets = [torch.tanh(torch.randn(B, 128, Hq >> i, Wq >> i, device=dev)) for i in range(3)]
    inps = [[torch.randn(B, 128, Hq >> i, Wq >> i, device=dev) for _ in range(3)] for i in range(3)]
    with torch.no_grad():
        for use_graph in (False, True):
            a = loop(f1, f2, [n.clone() for n in nets], inps, iters=2, test_mode=True, use_graph=use_graph)[1]
            loop.update_block.gru08.convq.weight.data.mul_(0.5)
            b = loop(f1, f2, [n.clone() for n in nets], inps, iters=2, test_mode=True, use_graph=use_graph)[1]
            real = C.supported, C.small_supported
            C.supported = C.small_supported = lambda *a, **k: False
            try:
                r = loop(f1, f2, [n.clone() for n in nets], inps, iters=2, test_mode=True)[1]
            finally:
                C.supported, C.small_supported = real
            assert not torch.allclose(a, b, atol=1e-4)
            assert float((b - r).abs().max()) <= 1e-3, float((b - r).abs().max())


@pytest.mark.parametrize("shape", [(1, 2, 3), (3, 1, 1), (1, 4, 16), (2, 3, 17), (0, 4, 8)])
def test_conv_nhwc_tiny_and_empty_images(S, shape):
    """Images smaller than one M block, a single pixel, an empty batch: every edge mode of the tile loop."""
    from stereoformer_amd import conv_split as C
    B, H, W = shape
    torch.manual_seed(89)
    x = torch.randn(B, 32, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    if not x.is_contiguous(memory_format=torch.channels_last) or x.is_contiguous():
        x = torch.empty_strided((B, 32, H, W), (32 * H * W, 1, 32 * W, 32), device="cuda").copy_(torch.randn(B, 32, H, W, device="cuda"))
    m = nn.Conv2d(32, 48, 3, padding=1).cuda()
    with torch.no_grad():
        if not C.supported([x], [m]):      # (1x1 images are both NCHW- and NHWC-contiguous: torch cannot tell them apart)
            pytest.skip("layout-ambiguous shape")
        (o,) = C.conv_nhwc([x], [m])
        r = m(x)
    assert o.shape == r.shape
    if B:
        assert float((o - r).abs().max()) <= 2e-5 * max(1.0, float(r.abs().max()))


@pytest.mark.parametrize("rows", [0, 2, 4])
def test_conv_nhwc_random_shapes_sweep(S, rows):
    """40 random (batch, size, source split, output split, kernel, ReLU, bias) combinations against the library's fp32 result,
    with the pixel-tile shape picked by launch size and forced to 2 / 4 rows."""
    from stereoformer_amd import conv_split as C
    if rows:
        with S._native.test_options(conv_rows=rows):
            return test_conv_nhwc_random_shapes_sweep(S, 0)
    g = torch.Generator().manual_seed(90)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for case in range(40):
        B, H, W = ri(1, 3), ri(1, 21), ri(1, 75)
        cs = tuple(16 * ri(1, 4) for _ in range(ri(1, 4)))
        nout = ri(1, 3)
        co = tuple([128 * ri(1, 2 if nout < 3 else 1) for _ in range(nout - 1)] + [ri(1, 160)])
        if sum(co) > 512:
            co = co[-1:]
        k = 3 if ri(0, 3) else 1
        relu, bias = bool(ri(0, 1)), bool(ri(0, 1))
        srcs = [torch.empty_strided((B, c, H, W), (c * H * W, 1, c * W, c), device="cuda").copy_(torch.randn(B, c, H, W, device="cuda"))
                for c in cs]
        convs = [nn.Conv2d(sum(cs), c, k, padding=k // 2, bias=bias).cuda() for c in co]
        with torch.no_grad():
            if not C.supported(srcs, convs):
                continue                      # (layout-ambiguous 1 x 1 images)
            outs = C.conv_nhwc(srcs, convs, relu=relu)
            x = torch.cat(srcs, 1)
            for o, m in zip(outs, convs):
                r = F.relu(m(x)) if relu else m(x)
                assert o.shape == r.shape, (case, B, H, W, cs, co, k)
                assert float((o - r).abs().max()) <= 3e-5 * max(1.0, float(r.abs().max())), (case, B, H, W, cs, co, k, relu, bias)


@pytest.mark.parametrize("flags", [dict(iter08=True, iter16=False, iter32=False), dict(iter08=True, iter16=True, iter32=False),
                                   dict(iter08=True, iter16=True, iter32=True, update=False), dict(iter08=False, iter16=True, iter32=True),
                                   dict(iter08=True, iter16=False, iter32=False, update=False)])
def test_update_block_partial_iterations_and_early_return(S, monkeypatch, flags):
    """The reference calls the block with subsets of its GRU levels (slow-fast schedule, raft_stereo.py:166-170) and with
    update=False (:170): with iter16 = iter32 = False the motion encoder is NOT forked to the side stream, with update=False the
    block returns right after the join.  Every combination under channels_last (conv_split + side stream) against the library
    path, twice in a row (a missed join would show up in the second call), eager and inside a HIP-graph capture."""
    from stereoformer_amd import conv_split as C
    torch.manual_seed(97)
    B, H, W = 1, 16, 32
    blk = S.BasicMultiUpdateBlock_LZH(4, 4, 3, 2, [128, 128, 128]).cuda().eval().to(memory_format=torch.channels_last)
    mk = lambda c, h, w: _cl(torch.randn(B, c, h, w, device="cuda") * 0.5)
    nets0 = [torch.tanh(mk(128, H >> i, W >> i)) for i in range(3)]
    inps = [[mk(128, H >> i, W >> i) for _ in range(3)] for i in range(3)]
    corr, flow = torch.randn(B, 36, H, W, device="cuda"), torch.randn(B, 2, H, W, device="cuda")

    def run():
        nets = [n.clone() for n in nets0]
        outs = []
        with torch.no_grad():
            for it in range(2):
                r = blk(nets, inps, corr, flow + 0.1 * it, **flags)
                nets = r if flags.get("update", True) is False else r[0]
                outs.append(r)
        flat = []
        for r in outs:
            for x in (r if flags.get("update", True) is False else [*r[0], r[1], r[2]]):
                flat.append(x.clone())
        return flat

    a = run()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        c = run()
    g.replay()
    torch.cuda.synchronize()
    monkeypatch.setattr(C, "supported", lambda *a, **k: False)
    monkeypatch.setattr(C, "small_supported", lambda *a, **k: False)
    b = run()
    for x, y, z in zip(a, b, c):
        assert x.shape == y.shape
        assert float((x - y).abs().max()) <= 2e-5 * max(1.0, float(y.abs().max())), float((x - y).abs().max())
        # (replay vs eager: to rounding, not bit for bit -- the library picks its 1x1 / 7x7 solvers per call)
        assert float((x - z).abs().max()) <= 2e-5 * max(1.0, float(x.abs().max())), float((x - z).abs().max())


def test_conv_nhwc_folds_batchnorm_and_pads_input_channels(S):
    """BasicGuidanceNet's encoders (Attempts/update.py:57-66: Conv2d(bias=False) -> BatchNorm2d -> ReLU over 24 / 3 channels) as
    ONE conv_split launch each: eval-mode BatchNorm folded into the packed weights, ReLU in the epilogue, the source zero-padded
    to a multiple of 16 channels -- against the fp64 conv -> bn -> relu of the unpadded input; the fold follows later updates
    of the BatchNorm statistics (refresh)."""
    from stereoformer_amd import conv_split as C
    torch.manual_seed(98)
    for cin, cpad in ((24, 32), (3, 16)):
        conv = nn.Conv2d(cin, 32, 3, padding=1, bias=False).cuda()
        bn = nn.BatchNorm2d(32).cuda().eval()
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.5)
            bn.running_var.uniform_(0.5, 2.0)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.3)
            x = torch.randn(2, cin, 11, 37, device="cuda")
            xp = torch.zeros(2, cpad, 11, 37, device="cuda").contiguous(memory_format=torch.channels_last)
            xp[:, :cin] = x
            assert not C.supported([xp], [conv]) and C.supported([xp], [conv], pad_in=True)
            (o,) = C.conv_nhwc([xp], [conv], relu=True, bns=[bn], pad_in=True)

            def ref():
                y = F.conv2d(x.double().cpu(), conv.weight.double().cpu(), None, padding=1)
                y = F.batch_norm(y, bn.running_mean.double().cpu(), bn.running_var.double().cpu(), bn.weight.double().cpu(),
                                 bn.bias.double().cpu(), False, 0.0, bn.eps)
                return F.relu(y)
            r = ref()
            assert float((o.double().cpu() - r).abs().max()) <= 4e-6 * float(r.abs().max()) + 1e-6
            bn.running_mean.data.add_(0.25)                       # (through .data: only the content fingerprint sees it)
            assert C.refresh([conv, bn]) == 1
            (o2,) = C.conv_nhwc([xp], [conv], relu=True, bns=[bn], pad_in=True)
            r2 = ref()
            assert float((o2.double().cpu() - r2).abs().max()) <= 4e-6 * float(r2.abs().max()) + 1e-6
            assert not torch.allclose(o, o2)
            bn.train()
            with pytest.raises(ValueError):
                C.conv_nhwc([xp], [nn.Conv2d(cin, 32, 3, padding=1, bias=False).cuda()], bns=[bn], pad_in=True)
            bn.eval()


def test_guidance_inputs_nhwc_matches_the_separate_kernels(S):
    """stereo_guidance_inputs_nhwc_f32 == (disp_warp(right, disp) - left, uncertainty_volume(vol, disp)) bit for bit (shared device
    functions), channels-last, zero in the padding channels."""
    from stereoformer_amd.update_dlc import guidance_inputs_nhwc
    torch.manual_seed(99)
    B, D, H, W = 2, 24, 9, 23
    vol = torch.randn(B, D, H, W, device="cuda") * 2
    disp = torch.rand(B, 1, H, W, device="cuda") * 25 - 1
    left, right = torch.randn(B, 3, H, W, device="cuda"), torch.randn(B, 3, H, W, device="cuda")
    with torch.no_grad():
        err, unc = guidance_inputs_nhwc(vol, disp, left, right)
        warped, _ = S.disp_warp(right, disp)
        u = S.uncertainty_volume(vol, disp)
    assert err.shape == (B, 16, H, W) and unc.shape == (B, 32, H, W) and err.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(err[:, :3], warped - left) and float(err[:, 3:].abs().max()) == 0.0
    assert torch.equal(unc[:, :D], u) and float(unc[:, D:].abs().max()) == 0.0


@pytest.mark.parametrize("case", [(2, 2, 64, 7, 13, 37), (1, 2, 64, 7, 8, 32), (1, 3, 32, 3, 5, 40), (2, 1, 16, 5, 9, 9), (1, 4, 48, 7, 3, 70)])
def test_conv_small_cin(S, case):
    """convf1 (7x7 over the 2-channel flow -> 64 channels, update.py:79,87) on the fp32 vector units: NCHW in, channels-last out,
    ReLU inside; against fp64."""
    from stereoformer_amd import conv_split as C
    B, cin, cout, k, H, W = case
    torch.manual_seed(100)
    x = torch.randn(B, cin, H, W, device="cuda") * 3
    m = nn.Conv2d(cin, cout, k, padding=k // 2).cuda()
    with torch.no_grad():
        assert C.small_cin_supported(x, m) and not C.small_cin_supported(x, nn.Conv2d(cin, cout + 1, k, padding=k // 2).cuda())
        for relu in (False, True):
            o = C.conv_small_cin(x, m, relu=relu)
            r = F.conv2d(x.double().cpu(), m.weight.double().cpu(), m.bias.double().cpu(), padding=k // 2)
            r = F.relu(r) if relu else r
            assert o.shape == r.shape and o.is_contiguous(memory_format=torch.channels_last)
            assert float((o.double().cpu() - r).abs().max()) <= 3e-6 * float(r.abs().max()) + 1e-6


def test_corr_lookup_channels_last_padded(S):
    """CorrBlock1D.__call__(coords, nhwc_channels=48): the NCHW lookup's values, channels-last, zero in the padding channels."""
    torch.manual_seed(101)
    B, C, H, W = 2, 64, 6, 48
    f1, f2 = torch.randn(B, C, H, W, device="cuda"), torch.randn(B, C, H, W, device="cuda")
    blk = S.CorrBlock1D(f1, f2, radius=4, num_levels=4)
    coords = S.coords_grid(B, H, W, device="cuda") + torch.randn(B, 2, H, W, device="cuda") * 6
    with torch.no_grad():
        a = blk(coords)
        b = blk(coords, nhwc_channels=48)
    assert b.shape == (B, 48, H, W) and b.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(b[:, :36], a) and float(b[:, 36:].abs().max()) == 0.0
    blk3 = S.CorrBlock1D(f1, f2, radius=3, num_levels=3)          # (the generic strided kernel)
    with torch.no_grad():
        a3, b3 = blk3(coords), blk3(coords, nhwc_channels=32)
    assert torch.equal(b3[:, :21], a3) and float(b3[:, 21:].abs().max()) == 0.0


def test_conv_nhwc_one_by_one_with_scale_and_640_channels(S):
    """mask[2] (update.py:149-152,176: 1x1, 256 -> 576, times 0.25) as one launch with the factor folded into the weights."""
    from stereoformer_amd import conv_split as C
    torch.manual_seed(102)
    x = _cl(torch.randn(1, 256, 9, 40, device="cuda"))
    m = nn.Conv2d(256, 576, 1).cuda()
    with torch.no_grad():
        assert C.supported([x], [m])
        (o,) = C.conv_nhwc([x], [m], scales=[0.25])
    r = 0.25 * F.conv2d(x.double().cpu(), m.weight.double().cpu(), m.bias.double().cpu())
    assert float((o.double().cpu() - r).abs().max()) <= 4e-6 * float(r.abs().max()) + 1e-6


@pytest.mark.parametrize("shape", [(2, 128, 17, 30), (1, 8, 5, 7), (1, 32, 136, 240), (2, 4, 2, 3)])
def test_pool2x_pool4x_channels_last(S, shape):
    """pool2x / pool4x (update.py:119-123) on channels-last activations: the library's kernel against F.avg_pool2d."""
    torch.manual_seed(103)
    x = torch.randn(*shape, device="cuda")
    xc = torch.empty_strided(shape, (shape[1] * shape[2] * shape[3], 1, shape[1] * shape[3], shape[1]), device="cuda").copy_(x)
    for fn, k, st in ((S.pool2x, 3, 2), (S.pool4x, 5, 4)):
        if shape[2] + 2 < k or shape[3] + 2 < k:
            continue
        with torch.no_grad():
            a = fn(xc)
        b = F.avg_pool2d(x, k, stride=st, padding=1)
        assert a.shape == b.shape and a.is_contiguous(memory_format=torch.channels_last)
        assert float((a - b).abs().max()) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 128, 17, 24), (1, 32, 3, 4), (3, 64, 5, 36), (1, 96, 34, 60)])
def test_relayout_nchw_to_nhwc(shape):
    """_native.nhwc: the tiled NCHW -> channels-last kernel (what the RAFT loop uses for the hidden states and context
    tensors the extractors hand over) is a pure permutation: bit-identical to Tensor.contiguous(memory_format=...); shapes it
    does not take (C % 32, H W % 4) and tensors that are already channels-last go the torch way."""
    from stereoformer_amd import _native as N
    torch.manual_seed(5)
    x = torch.randn(*shape, device="cuda")
    y = N.nhwc(x)
    assert y.is_contiguous(memory_format=torch.channels_last) and y.shape == x.shape
    assert torch.equal(y, x)
    assert torch.equal(y.permute(0, 2, 3, 1).contiguous(), x.permute(0, 2, 3, 1).contiguous())
    assert N.nhwc(y) is y or N.nhwc(y).data_ptr() == y.data_ptr()
    odd = torch.randn(1, 24, 3, 5, device="cuda")
    z = N.nhwc(odd)
    assert z.is_contiguous(memory_format=torch.channels_last) and torch.equal(z, odd)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 1, 9, 70, 8), (1, 2, 5, 130, 4), (1, 1, 3, 64, 4), (2, 2, 4, 33, 8)])
def test_upsample_convex_reads_a_channels_last_mask(cfg):
    """upsample_convex with the mask channels-last (what the update blocks' mask heads hand over from the channels-last
    convolution kernels): same values as through the NCHW mask, no copy of the mask in between."""
    import stereoformer_amd as S
    n, cd, h, w, f = cfg
    torch.manual_seed(11)
    disp = torch.randn(n, cd, h, w, device="cuda") * 5
    mask = torch.randn(n, 9 * f * f, h, w, device="cuda")
    ref = S.upsample_convex(disp, mask, f)
    got = S.upsample_convex(disp, mask.contiguous(memory_format=torch.channels_last), f)
    assert got.shape == ref.shape and torch.equal(got, ref)
