"""GPU parity of the update block's convolutions on the split-bf16 matrix-core kernel (csrc/conv_split.hip) against the fp64
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
def test_conv_nhwc_matches_fp64_and_miopen(S, case):
    from stereoformer_amd import conv_split as C
    B, H, W, cs, co, k, relu, bias = case
    torch.manual_seed(81)
    srcs = [_cl(torch.randn(B, c, H, W, device="cuda")) for c in cs]
    convs = [nn.Conv2d(sum(cs), c, k, padding=k // 2, bias=bias).cuda() for c in co]
    assert C.supported(srcs, convs)
    with torch.no_grad():
        outs = C.conv_nhwc(srcs, convs, relu=relu)
        lib = [m(torch.cat(srcs, 1)) for m in convs]
    ref = _ref64(srcs, convs, relu)
    for o, r, l, m in zip(outs, ref, lib, convs):
        assert o.shape == r.shape and o.is_contiguous(memory_format=torch.channels_last)
        scale = float(r.abs().max())
        err = float((o.double().cpu() - r).abs().max())
        err_lib = float(((F.relu(l) if relu else l).double().cpu() - r).abs().max())
        # fp32-faithful: within a few ulps of the fp64 result at the output's scale (six fp32 accumulations per 16 input
        # channels instead of the f32 instruction's eight: the same order of rounding error as the library's own result)
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
    assert not C.supported([x.requires_grad_()], [m])                               # training keeps the library path
    with pytest.raises(ValueError):
        C.conv_nhwc([x.detach().contiguous()], [m])
