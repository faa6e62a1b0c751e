"""dev: the headline launch timed with its 25.5-GB volume buffer as a WINDOW of one large allocation, at offsets 4 GB apart:
is the fast / slow property finer-grained than a whole allocation?  python tools/place_window_probe.py [total_GB]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
tot = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
B, C, H, W, D, G = 8, 32, 540, 960, 192, 8
L = torch.randn(B, C, H, W, device=dev); R = torch.randn(B, C, H, W, device=dev)
n = B * G * D * H * W
big = torch.empty(tot * (1 << 30) // 4, device=dev, dtype=torch.float32)
disp = torch.empty((B, H, W), device=dev); mean = torch.empty((B, D, H, W), device=dev)


def t(vol, reps=2):
    with torch.no_grad():
        S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=(vol, disp, mean))
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=(vol, disp, mean)); e1.record()
            torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best


step = (2 << 30) // 4
res = []
off = 0
while off + n <= big.numel():
    res.append((off * 4 / 2**30, t(big[off:off + n].view(B, G, D, H, W))))
    off += step
print(" ".join("%.0f:%.2f" % r for r in res))
print("min %.3f max %.3f" % (min(r[1] for r in res), max(r[1] for r in res)))

# can the best window be re-created as its own allocation?  free the arena, allocate [padding of the best offset][volume]
best_off = min(res, key=lambda r: r[1])[0]
worst_off = max(res, key=lambda r: r[1])[0]
del big
torch.cuda.empty_cache()
for name, off_gb in (("best", best_off), ("worst", worst_off), ("best", best_off)):
    pad = torch.empty(int(off_gb * (1 << 30)) // 4, device=dev, dtype=torch.float32) if off_gb > 0 else None
    vol = torch.empty((B, G, D, H, W), device=dev)
    print("re-created at the %s offset (%.0f GB of padding in front): %.3f ms" % (name, off_gb, t(vol)))
    del pad, vol
    torch.cuda.empty_cache()
vol = torch.empty((B, G, D, H, W), device=dev)
print("plain allocation: %.3f ms" % t(vol))
