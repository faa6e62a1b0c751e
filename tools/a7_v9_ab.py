"""dev: cfg-3 cross-attention, split-fp16 (v9, default) vs split-bf16 (v7) vs fp32 online (v5), fp32 and fp16 features, and the
workgroup shapes of v9.   python tools/a7_v9_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from tools.kernel_bench import timeit
B, C, H, W, D = 8, 32, 540, 960, 192
L = torch.randn(B, C, H, W, device="cuda"); R = torch.randn(B, C, H, W, device="cuda")
att = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")
for half in (False, True):
    a, b = (L.half(), R.half()) if half else (L, R)
    for opts in ({"att_kernel": 0}, {"att_kernel": 0, "att_waves": 6}, {"att_kernel": 0, "att_waves": 8}, {"att_kernel": 7}, {"att_kernel": 5}):
        with S._native.test_options(**opts), torch.no_grad():
            ms = timeit(lambda: att(a, b), 20, 5)
            ms2 = timeit(lambda: att(a, b, need_out=False), 20, 5)
        print(f"{'fp16' if half else 'fp32'} features, {opts}: out + disparity {ms:.3f} ms, disparity only {ms2:.3f} ms", flush=True)
