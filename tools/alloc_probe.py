"""Probe behind stereoformer_amd/hbm.py: does the fused volume-stage launch time depend on WHERE its 28.7 GB of outputs live?
Seven arenas are held at once (different physical regions of the 288 GB), each is timed with the launch and with a plain
linear fill, three are re-timed; then everything is freed and re-allocated (same virtual addresses, new physical memory).
    python tools/alloc_probe.py > profiles/rNN_alloc_probe.txt"""
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


def run(arena, reps=6):
    vol = arena[:nvol]
    mean = arena[nvol + 4096:nvol + 4096 + nmean]
    disp = arena[nvol + nmean + 8192:nvol + nmean + 8192 + ndisp]

    def once():
        N.call("stereo_gwc_volume_soft_argmin_f32", N.ptr(L), N.ptr(R), N.ptr(vol), N.ptr(mean), N.ptr(disp), B, C, H, W, D, G, 1,
               N.stream_ptr())
    once()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); once(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), max(ts)


def fill_ms(arena, reps=3):
    """A plain linear write stream over the same bytes (torch fill kernel)."""
    arena.fill_(1.0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); arena.fill_(1.0); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


for rnd in range(2):
    arenas = [torch.empty(nvol + nmean + ndisp + (1 << 20), device="cuda") for _ in range(7)]
    for i, a in enumerate(arenas):
        mn, mx = run(a)
        fm = fill_ms(a)
        print(f"round {rnd} arena {i} @{a.data_ptr():#x}: fused volume stage {mn:.3f} .. {mx:.3f} ms over 6 launches = "
              f"{29743718400 / mn / 1e6:.0f} GB/s algorithmic | linear fill of the same arena {fm:.3f} ms = {a.numel() * 4 / fm / 1e6:.0f} GB/s",
              flush=True)
    for i in (3, 0, 5):   # re-time a few: is the figure a property of the arena?
        mn, mx = run(arenas[i])
        print(f"round {rnd} arena {i} again: {mn:.3f} .. {mx:.3f} ms", flush=True)
    del arenas, a
    torch.cuda.empty_cache()
