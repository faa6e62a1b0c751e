"""dev: the split-operand convolution kernel (both fp32-faithful forms) vs MIOpen fp32 at the update block's shapes, and the
ConvGRU with its gate math in the epilogues vs linear epilogues + pointwise kernels.   python tools/conv_ab.py"""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import conv_split as C
from stereoformer_amd.update_dlc import gru_blend, gru_reset
from tools.kernel_bench import timeit
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for (nm, B, H, W, cs, co, relu) in (("gru08 convz|convr [8,384,136,240] -> 128+128", 8, 136, 240, (128, 128, 128), (128, 128), False),
                                    ("gru08 convq 384 -> 128", 8, 136, 240, (128, 128, 128), (128,), False),
                                    ("flow_head.conv1|mask[0] 128 -> 256+256", 8, 136, 240, (128,), (256, 256), True),
                                    ("encoder.conv 128 -> 126", 8, 136, 240, (64, 64), (126,), True),
                                    ("encoder.convc2 64 -> 64", 8, 136, 240, (64,), (64,), True),
                                    ("gru16 convz|convr [8,384,68,120]", 8, 68, 120, (128, 128, 128), (128, 128), False),
                                    ("gru32 convz|convr [8,256,34,60]", 8, 34, 60, (128, 128), (128, 128), False)):
    srcs = [cl(torch.randn(B, c, H, W, device="cuda")) for c in cs]
    convs = [nn.Conv2d(sum(cs), c, 3, padding=1).cuda().to(memory_format=torch.channels_last) for c in co]
    gf = 2 * B * H * W * sum(cs) * sum(co) * 9 / 1e9
    with torch.no_grad():
        ms = timeit(lambda: C.conv_nhwc(srcs, convs, relu=relu, precision="f16x3"), 10, 3)
        mss = timeit(lambda: C.conv_nhwc(srcs, convs, relu=relu, precision="f16x3s"), 10, 3)
        ms6 = timeit(lambda: C.conv_nhwc(srcs, convs, relu=relu, precision="bf16x6"), 10, 3)
        x = torch.cat(srcs, 1)
        ms_lib = timeit(lambda: [m(x) for m in convs], 10, 3)
    print(f"{nm}: split-fp16 (3 products) 4x1 tile, two accumulators {ms:.3f} ms = {gf / ms:.0f} TFLOP/s fp32-equivalent | 4x2 tile, one "
          f"accumulator {mss:.3f} ms = {gf / mss:.0f} | split-bf16 (6 products) {ms6:.3f} ms = "
          f"{gf / ms6:.0f} | MIOpen convs alone {ms_lib:.3f} ms = {gf / ms_lib:.0f}", flush=True)

# the whole ConvGRU step at 1/4 scale
B, H, W = 8, 136, 240
h = cl(torch.tanh(torch.randn(B, 128, H, W, device="cuda")))
xs = [cl(torch.randn(B, 128, H, W, device="cuda")) for _ in range(2)]
cz, cr, cq = [cl(torch.randn(B, 128, H, W, device="cuda")) for _ in range(3)]
convz, convr, convq = [nn.Conv2d(384, 128, 3, padding=1).cuda().to(memory_format=torch.channels_last) for _ in range(3)]


def fused(p):
    z, g = C.conv_nhwc([h, *xs], [convz, convr], epilogue=[("sigmoid", cz), ("sigmoid_mul", cr, h)], precision=p)
    return C.conv_nhwc([g, *xs], [convq], epilogue=[("gru_blend", cq, h, z)], precision=p)[0]


def unfused(p):
    zp, rp = C.conv_nhwc([h, *xs], [convz, convr], precision=p)
    (qp,) = C.conv_nhwc([gru_reset(h, rp, cr), *xs], [convq], precision=p)
    return gru_blend(h, zp, qp, cz, cq)


with torch.no_grad():
    for p in ("f16x3", "f16x3s", "bf16x6"):
        a, b = timeit(lambda: fused(p), 10, 3), timeit(lambda: unfused(p), 10, 3)
        print(f"ConvGRU step [8,128+256,136,240], precision {p}: gate math in the epilogues {a:.3f} ms | linear epilogues + gru_reset + gru_blend {b:.3f} ms",
              flush=True)

x = cl(torch.randn(8, 256, 136, 240, device="cuda"))
m = nn.Conv2d(256, 2, 3, padding=1).cuda().to(memory_format=torch.channels_last)
with torch.no_grad():
    a = timeit(lambda: C.conv3x3_small(x, m), 10, 3)
    b = timeit(lambda: m(x), 10, 3)
print(f"flow_head.conv2 256 -> 2: fp32 vector kernel {a:.3f} ms | MIOpen {b:.3f} ms", flush=True)
