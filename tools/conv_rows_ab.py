"""dev: 2-row vs 4-row pixel tiles of the split-fp16 convolution kernel at the small / medium launches of the loops (test hook
conv_rows).   python tools/conv_rows_ab.py"""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import conv_split as C
from tools.kernel_bench import timeit
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for (nm, B, H, W, cs, co) in (("cfg-4 gru16 z|r", 8, 68, 120, (128, 128, 128), (128, 128)), ("cfg-4 gru16 q", 8, 68, 120, (128, 128, 128), (128,)),
                              ("cfg-4 gru32 z|r", 8, 34, 60, (128, 128), (128, 128)), ("cfg-4 gru32 q", 8, 34, 60, (128, 128), (128,)),
                              ("LowCNN z|b", 8, 72, 120, (32, 32, 64), (64, 64)), ("LowCNN g", 8, 72, 120, (64, 32, 32), (64,)),
                              ("LowCNN offset.conv1", 8, 72, 120, (64,), (64,)), ("LowCNN enc 32->32", 8, 72, 120, (32,), (32,)),
                              ("cfg-5 gru08 z|r", 4, 96, 312, (128, 128, 128), (128, 128)), ("cfg-5 gru08 q", 4, 96, 312, (128, 128, 128), (128,)),
                              ("cfg-5 gru16 z|r", 4, 48, 156, (128, 128, 128), (128, 128)), ("cfg-5 head", 4, 96, 312, (128,), (256,)),
                              ("cfg-4 encoder 64->64", 8, 136, 240, (64,), (64,)), ("cfg-4 encoder 128->126", 8, 136, 240, (64, 64), (126,)),
                              ("cfg-4 head 128->256", 8, 136, 240, (128,), (256,)), ("cfg-4 gru08 q", 8, 136, 240, (128, 128, 128), (128,))):
    srcs = [cl(torch.randn(B, c, H, W, device="cuda")) for c in cs]
    convs = [nn.Conv2d(sum(cs), c, 3, padding=1).cuda() for c in co]
    tx, ntn = (W + 31) // 32, (sum(co) + 127) // 128
    res = []
    with torch.no_grad():
        for rows in (2, 4):
            with S._native.test_options(conv_rows=rows):
                res.append(timeit(lambda: C.conv_nhwc(srcs, convs), 20, 5))
    print(f"{nm}: 4-row workgroup tiles {tx * ((H + 3) // 4) * B * ntn}: 2 rows {res[0] * 1e3:.1f} us | 4 rows {res[1] * 1e3:.1f} us", flush=True)
