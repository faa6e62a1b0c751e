"""dev: the cfg-3 attention step timed over 5 / 20 / 100 / 300 back-to-back calls (clock ramp of an idle GPU vs steady state).
python tools/a7_sustained.py"""
import os, sys, torch, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
dev = torch.device("cuda")
B, C, H, W, D = 8, 32, 540, 960, 192
L = torch.randn(B, C, H, W, device=dev); R = torch.randn(B, C, H, W, device=dev)
att = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")
for _ in range(3): att(L, R)
torch.cuda.synchronize()
for n in (5, 20, 100, 300):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): att(L, R)
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(n, "reps: event %.3f ms/call, wall %.3f ms/call" % (e0.elapsed_time(e1) / n, (t1 - t0) / n * 1e3))
