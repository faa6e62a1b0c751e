"""dev: A1 backward (dL + dR) at cfg-2 size under the test options given as k=v arguments (corr_bwd_fused = 0: round 2's two
launches, 1: the one-read kernel, 2 / 3: its timing ablations without arithmetic / without volume loads).
python tools/a1bwd_ab.py corr_bwd_fused=2 [B=2]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import _native
dev = torch.device("cuda")
B, C, H, W, D = 8, 32, 540, 960, 192
opts = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a and not a.startswith("B="))
for a in sys.argv[1:]:
    if a.startswith("B="):
        B = int(a[2:])
L = torch.randn(B, C, H, W, device=dev)
R = torch.randn(B, C, H, W, device=dev)
g = torch.randn(B, D, H, W, device=dev)
gl, gr = torch.empty_like(L), torch.empty_like(R)
lib = _native.load()
def run():
    _native.check(lib.stereo_gwc_volume_bwd_f32(_native.ptr(L), _native.ptr(R), _native.ptr(g), _native.ptr(gl), _native.ptr(gr),
                                                B, C, H, W, D, 1, _native.stream_ptr()), "bwd")
with _native.test_options(**opts):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
gb = (B * D * H * W * 4 + 4 * B * C * H * W * 4) / 1e9
print(f"{opts} B={B}: {ms:.3f} ms  {gb / ms:.2f} TB/s of {gb:.2f} GB (one read of the volume + features in / out)")

