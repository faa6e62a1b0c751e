"""dev: the cfg-5 disparity-only attention launch ([4,256,96,312], D=64, channel-chunked kernel) under a few workgroup shapes.
python tools/a7_cfg5_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import _native as N
B, C, H, W, D = 4, 256, 96, 312, 64
L = torch.randn(B, C, H, W, device="cuda"); R = torch.randn(B, C, H, W, device="cuda")
att = S.CrossAttentionCostVolume(D, "neginf", "sqrt_c")
gb = (2 * B * C * H * W * 4 + B * H * W * 4) / 1e9
with torch.no_grad():
    ref = att(L, R, need_out=False)
for opts in ({}, {"att_waves": 4}, {"att_waves": 5}, {"att_waves": 8}, {"att_kernel": 17}, {"att_kernel": 17, "att_waves": 4},
             {"att_kernel": 17, "att_waves": 5}, {"att_kernel": 17, "att_waves": 8}, {"att_waves": 4, "tiles_per_wg": 1},
             {"att_kernel": 17, "att_waves": 4, "tiles_per_wg": 1}, {"att_kernel": 17, "att_waves": 5, "tiles_per_wg": 1}):
    with N.test_options(**opts), torch.no_grad():
        got = att(L, R, need_out=False)
        got = got[1] if isinstance(got, tuple) else got
        r_ = ref[1] if isinstance(ref, tuple) else ref
        print("max |diff to default| %.2e" % float((got - r_).abs().max()), end="  ")
        for _ in range(30): att(L, R, need_out=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): att(L, R, need_out=False)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
    print(f"{opts}: {ms * 1e3:.1f} us  {gb / ms:.2f} TB/s = {gb / ms / 8:.3f} of HBM")
