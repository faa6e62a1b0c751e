"""dev: the fused volume-stage launch on seven output arenas held at once, for a rocprofv3 --pmc pass with
PER-INSTANCE L2-channel write counters -- is a slow arena one whose writes pile up on few channels?
    rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL -d out -o run --output-format csv -- python3 tools/arena_pmc.py
Dispatch order of corr_volume_mfma_kernel: arena 0 x3, arena 1 x3, ... (the first launch of each is the warm-up)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S  # noqa: E402
from stereoformer_amd import _native as N  # noqa: E402

B, C, H, W, D, G = 8, 32, 540, 960, 192, 8
L = torch.randn(B, C, H, W, device="cuda")
R = torch.randn(B, C, H, W, device="cuda")
nvol, nmean, ndisp = B * G * D * H * W, B * D * H * W, B * H * W
arenas = [torch.empty(nvol + nmean + ndisp + (1 << 20), device="cuda") for _ in range(7)]
for i, arena in enumerate(arenas):
    vol = arena[:nvol]
    mean = arena[nvol + 4096:nvol + 4096 + nmean]
    disp = arena[nvol + nmean + 8192:nvol + nmean + 8192 + ndisp]
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        N.call("stereo_gwc_volume_soft_argmin_f32", N.ptr(L), N.ptr(R), N.ptr(vol), N.ptr(mean), N.ptr(disp), B, C, H, W, D, G, 1,
               N.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"arena {i} @{arena.data_ptr():#x}: {ts[1]:.3f} {ts[2]:.3f} ms", flush=True)
