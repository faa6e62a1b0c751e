"""dev: a few cfg-4 LowCNN-GRU loop passes (eager) for `rocprofv3 --kernel-trace --stats`.   python tools/lowcnn_trace.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
dev = torch.device("cuda")
B = 8
g = torch.Generator(device=dev).manual_seed(450)
vol = torch.randn(B, 24, 72, 120, device=dev, generator=g)
disp = torch.rand(B, 1, 72, 120, device=dev, generator=g) * 20
li = torch.randn(B, 3, 576, 960, device=dev, generator=g)
ri = torch.randn(B, 3, 576, 960, device=dev, generator=g)
torch.manual_seed(4)
lp = S.LowCNNGRULoop(24, 32, 20).eval().to(dev)
with torch.no_grad():
    for _ in range(3):
        lp(vol, disp, li, ri, iters=16, use_graph=False)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        lp(vol, disp, li, ri, iters=16, use_graph=True)
    torch.cuda.synchronize()
    print("graph replay ms:", (time.perf_counter() - t0) / 5 * 1e3)
