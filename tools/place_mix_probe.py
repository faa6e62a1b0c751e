"""dev: does the headline launch's speed depend on WHICH buffer sits in a slow arena?  Times build_gwc_volume_with_disparity for
every (volume buffer i, mean buffer j) combination of a few candidate allocations.  python tools/place_mix_probe.py [n]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda")
B, C, H, W, D, G = 8, 32, 540, 960, 192, 8
L = torch.randn(B, C, H, W, device=dev); R = torch.randn(B, C, H, W, device=dev)
cands = [(torch.empty((B, G, D, H, W), device=dev), torch.empty((B, H, W), device=dev), torch.empty((B, D, H, W), device=dev)) for _ in range(n)]


def t(bufs, reps=3):
    with torch.no_grad():
        S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=bufs)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=bufs); e1.record()
            torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best


print("rows: volume buffer i, columns: mean buffer j (ms)")
for i in range(n):
    print(" ".join("%.3f" % t((cands[i][0], cands[i][1], cands[j][2])) for j in range(n)))
