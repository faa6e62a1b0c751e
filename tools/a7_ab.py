"""A/B of the C > 32 cross-attention kernels (dev tool): fp32 two-pass (v4) vs chunked pipeline (v6) x workgroup shapes, fp32 and
fp16 features.   python tools/a7_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

for (nm, bb, cc, hh, ww, dd) in (("cfg-3b [8,256,136,240] D=48", 8, 256, 136, 240, 48), ("cfg-5 [4,256,96,312] D=64", 4, 256, 96, 312, 64),
                              ("[4,256,96,312] D=128", 4, 256, 96, 312, 128), ("[8,64,136,240] D=100", 8, 64, 136, 240, 100)):
    q1 = torch.randn(bb, cc, hh, ww, device="cuda")
    q2 = torch.randn(bb, cc, hh, ww, device="cuda")
    a2 = S.CrossAttentionCostVolume(dd, "neginf", "sqrt_c")
    for half in (False, True):
        x1, x2 = (q1.half(), q2.half()) if half else (q1, q2)
        for opts in ({"att_kernel": 4}, {"att_kernel": 6, "att_waves": 4}, {"att_kernel": 6, "att_waves": 5}, {"att_kernel": 6, "att_waves": 8},
                     {"att_kernel": 8, "att_waves": 4}, {"att_kernel": 8, "att_waves": 5}, {"att_kernel": 8, "att_waves": 8}):
            if half and opts["att_kernel"] == 4:
                continue
            with S._native.test_options(**opts), torch.no_grad():
                ms = timeit(lambda: a2(x1, x2), 30, 5)
                ms2 = timeit(lambda: a2(x1, x2, need_out=False), 30, 5)
            print(nm, "fp16" if half else "fp32", opts, round(ms, 4), round(ms2, 4), flush=True)
