"""dev: the cfg-5 disparity-only attention launch ([4,256,96,312], D=64) timed as a replayed graph of 50 launches (no host in
the loop; do NOT run this under rocprofv3 -- graph capture under the profiler hung).
python tools/a7_cfg5_trace.py [att_kernel ...]   (0 default, 8 split-bf16 form)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import _native as N
B, C, H, W, D = 4, 256, 96, 312, 64
L = torch.randn(B, C, H, W, device="cuda"); R = torch.randn(B, C, H, W, device="cuda")
att = S.CrossAttentionCostVolume(D, "neginf", "sqrt_c")
gb = (2 * B * C * H * W * 4 + B * H * W * 4) / 1e9
for k in [int(a) for a in sys.argv[1:]] or [0]:
    with N.test_options(att_kernel=k), torch.no_grad():
        for _ in range(30):
            att(L, R, need_out=False)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            att(L, R, need_out=False)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(50):
                    att(L, R, need_out=False)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 500 * 1e3
        print("att_kernel=%d: %.1f us per launch (graph of 50), %.3f of HBM" % (k, us, gb / (us * 1e-6) / 8e3), flush=True)
