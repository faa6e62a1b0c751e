"""dev: convf1 (7x7, 2 -> 64, + ReLU) on the fp32 vector kernel vs the library.   python tools/conv7_ab.py"""
import os, sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import conv_split as C
from tools.kernel_bench import timeit
for (B, H, W) in ((8, 136, 240), (4, 96, 312)):
    x = torch.randn(B, 2, H, W, device="cuda")
    m = nn.Conv2d(2, 64, 7, padding=3).cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():
        a = timeit(lambda: C.conv_small_cin(x, m, relu=True), 20, 5)
        b = timeit(lambda: F.relu(m(x)), 20, 5)
    print(f"convf1 [{B},2,{H},{W}] -> 64: vector kernel (+ ReLU, NHWC out) {a * 1e3:.1f} us | library conv + ReLU {b * 1e3:.1f} us", flush=True)
