import os, sys, torch
sys.path.insert(0, '/root/repo')
import stereoformer_amd as S
from tools.kernel_bench import timeit
B, C, H, W, D = 8, 32, 540, 960, 192
L = torch.randn(B, C, H, W, device="cuda"); R = torch.randn(B, C, H, W, device="cuda")
att = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")
for half in (False, True):
    l, r = (L.half(), R.half()) if half else (L, R)
    for opts in ({}, {"att_kernel": 7}, {"att_kernel": 7, "att_waves": 6}, {"att_kernel": 7, "att_waves": 8}, {"att_kernel": 7, "tiles_per_wg": 17}):
        with S._native.test_options(**opts), torch.no_grad():
            ms = timeit(lambda: att(l, r), 10, 2); ms2 = timeit(lambda: att(l, r, need_out=False), 10, 2)
        print("fp16" if half else "fp32", opts, round(ms, 4), round(ms2, 4), flush=True)
