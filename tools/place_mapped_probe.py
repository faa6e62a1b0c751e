"""dev: the headline launch with its volume as a window of a MAPPED arena (csrc/hbm_arena.hip): (1) a small functional check
of create / window / keep; (2) the window sweep of tools/place_window_probe.py on chunk-mapped memory; (3) does the best window
keep its speed after every other chunk has been released?   python tools/place_mapped_probe.py [span_GB] [chunk_MB]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import hbm
span_gb = int(sys.argv[1]) if len(sys.argv) > 1 else 160
chunk = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
dev = torch.device("cuda", 0)
print("granularity", hbm.MappedArena.granularity(dev), flush=True)

a = hbm.MappedArena(64 << 20, 8, dev)
w = a.window(0, 8 * (64 << 20))
w.fill_(3.0)
w[(64 << 20) // 4 * 3:(64 << 20) // 4 * 5].copy_(torch.arange((64 << 20) // 2, device=dev, dtype=torch.float32))
torch.cuda.synchronize()
del w
free0 = torch.cuda.mem_get_info(dev)[0]
a.keep(3, 2)
free1 = torch.cuda.mem_get_info(dev)[0]
w = a.window(3, 2 * (64 << 20))
assert torch.equal(w, torch.arange((64 << 20) // 2, device=dev, dtype=torch.float32)), "kept chunks lost their content"
print("keep released %d MiB; kept content intact" % ((free1 - free0) >> 20), flush=True)
try:
    a.window(0, 64 << 20)
    raise SystemExit("window outside the kept chunks was handed out")
except ValueError:
    pass
del w, a
print("free after destroy: +%d MiB" % ((torch.cuda.mem_get_info(dev)[0] - free1) >> 20), flush=True)

B, C, H, W, D, G = 8, 32, 540, 960, 192, 8
L = torch.randn(B, C, H, W, device=dev); R = torch.randn(B, C, H, W, device=dev)
n = B * G * D * H * W
disp = torch.empty((B, H, W), device=dev); mean = torch.empty((B, D, H, W), device=dev)


def t(vol, reps=2):
    vol = vol.view(B, G, D, H, W)
    with torch.no_grad():
        S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=(vol, disp, mean))
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=(vol, disp, mean)); e1.record()
            torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best


own = -(-4 * n // chunk)
nch = own + span_gb * (1 << 30) // chunk
t0 = time.time()
arena = hbm.MappedArena(chunk, nch, dev)
print("arena of %d chunks of %d MiB mapped in %.2f s" % (nch, chunk >> 20, time.time() - t0), flush=True)
step = (2 << 30) // chunk
res = []
k = 0
while k + own <= nch:
    res.append((k, t(arena.window(k, 4 * n))))
    k += step
print(" ".join("%.0f:%.2f" % (k * chunk / 2**30, ms) for k, ms in res))
print("min %.3f max %.3f" % (min(r[1] for r in res), max(r[1] for r in res)), flush=True)
kb = min(res, key=lambda r: r[1])[0]
t0 = time.time()
arena.keep(kb, own)
print("released everything outside the best window in %.2f s; free now %.1f GiB" % (time.time() - t0,
                                                                                   torch.cuda.mem_get_info(dev)[0] / 2**30))
vol = arena.window(kb, 4 * n)
print("best window after the release: %.3f %.3f %.3f ms" % (t(vol), t(vol), t(vol)), flush=True)
# other memory traffic in the process afterwards must not disturb it
other = [torch.empty(8 << 28, device=dev) for _ in range(4)]
for o in other:
    o.fill_(1.0)
print("... with 32 GB of other tensors allocated and written: %.3f ms" % t(vol))
del other
plain = torch.empty(n, device=dev)
print("plain allocation: %.3f ms" % t(plain))
del plain

win, rep = hbm.place_in_mapped_arena(4 * n, lambda w: t(w), span_bytes=span_gb << 30, chunk_bytes=chunk, device=dev)
print(rep)
