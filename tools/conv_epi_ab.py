"""dev: cost of the gate epilogues of conv_split (fp32 / fp16 activations) against linear epilogues.   python tools/conv_epi_ab.py"""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import conv_split as C
from tools.kernel_bench import timeit
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for dt in (torch.float32, torch.float16):
    for (B, H, W) in ((8, 136, 240), (8, 68, 120), (8, 34, 60)):
        h = cl(torch.tanh(torch.randn(B, 128, H, W, device="cuda"))).to(dt)
        xs = [cl(torch.randn(B, 128, H, W, device="cuda")).to(dt) for _ in range(2)]
        cz, cr, cq = [cl(torch.randn(B, 128, H, W, device="cuda")).to(dt) for _ in range(3)]
        convz, convr, convq = [nn.Conv2d(384, 128, 3, padding=1).cuda().to(memory_format=torch.channels_last) for _ in range(3)]
        with torch.no_grad():
            a = timeit(lambda: C.conv_nhwc([h, *xs], [convz, convr]), 10, 3)
            b = timeit(lambda: C.conv_nhwc([h, *xs], [convz, convr], epilogue=[("sigmoid", cz), ("sigmoid_mul", cr, h)]), 10, 3)
            z, g = C.conv_nhwc([h, *xs], [convz, convr], epilogue=[("sigmoid", cz), ("sigmoid_mul", cr, h)])
            c = timeit(lambda: C.conv_nhwc([g, *xs], [convq]), 10, 3)
            d = timeit(lambda: C.conv_nhwc([g, *xs], [convq], epilogue=[("gru_blend", cq, h, z)]), 10, 3)
        print(f"{dt} [{B},384,{H},{W}]: convz|convr linear {a:.3f} ms, gate epilogues {b:.3f} ms | convq linear {c:.3f} ms, blend epilogue {d:.3f} ms", flush=True)
