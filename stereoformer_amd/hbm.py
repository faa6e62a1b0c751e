"""HBM placement of large, long-lived output buffers.

Measured on MI355X (``tools/alloc_probe.py``): the fused volume-stage launch, which streams 28.7 GB of
row-contiguous 1-KiB stores into ~8 000 concurrent output rows, runs at one of two stable speeds depending on WHICH physical
memory its output buffers were given by the driver -- 4.65-4.9 ms (6.1-6.4 TB/s, the rate of a plain linear fill) or
5.4-5.9 ms -- and keeps that speed for as long as the buffers live; a linear fill of the same bytes runs at the same rate on
every arena, and no tile schedule tried removes the difference.  The speed is a property of the allocation, it is reproducible,
and the caching allocator reuses a freed block for the next step anyway.  So a caller that owns its output buffers for many
steps (inference service, training loop, the benchmark) allocates them through ``place``: a few candidate allocations are
timed with the launch that will use them, the fastest is kept, the others are returned to the driver.

Nothing here changes what is computed or where results are written: ``out=`` buffers are ordinary tensors that every step
overwrites completely.
"""
from __future__ import annotations

import torch


def place(alloc, probe, max_tries=6, good_enough_ms=None, reserve_bytes=8 << 30, bytes_per_try=None, device=None,
          max_hold_fraction=0.85, release_to_driver=True):
    """Allocate with ``alloc()`` up to ``max_tries`` times, time each candidate with ``probe(candidate) -> ms``, keep the
    fastest and release the rest.  Rejected candidates stay alive until the end (freeing one would hand the same physical
    block back to the next ``alloc``), so the search holds up to ``max_tries * bytes_per_try`` transiently: it stops early at a
    candidate within ``good_enough_ms``, when fewer than ``reserve_bytes`` + ``bytes_per_try`` of ``device``'s memory remain
    free, or when the candidates held would exceed ``max_hold_fraction`` of the device's total memory.
    ``release_to_driver``: finish with ``torch.cuda.empty_cache()`` so that the rejected blocks really leave the process --
    NOTE that this flushes the whole caching allocator of the process (every cached free block, not only ours); a service that
    shares the process with other GPU work can pass ``False`` and let the allocator keep the blocks for reuse.
    Returns ``(candidate, report)``."""
    held, times = [], []
    best = None
    dev = torch.device(device) if device is not None else None
    for _ in range(max(1, int(max_tries))):
        if held and bytes_per_try is not None and torch.cuda.is_available():
            free, total = torch.cuda.mem_get_info(dev)
            if free < bytes_per_try + reserve_bytes or (len(held) + 1) * bytes_per_try > max_hold_fraction * total:
                break
        cand = alloc()
        ms = float(probe(cand))
        held.append(cand)
        times.append(ms)
        if best is None or ms < times[best]:
            best = len(held) - 1
        if good_enough_ms is not None and ms <= good_enough_ms:
            break
    chosen = held[best]
    report = {"probe_ms": [round(t, 4) for t in times], "chosen": best, "chosen_ms": round(times[best], 4),
              "first_allocation_ms": round(times[0], 4),
              "peak_bytes_held": None if bytes_per_try is None else len(held) * int(bytes_per_try)}
    del held, cand
    if release_to_driver and torch.cuda.is_available():
        torch.cuda.empty_cache()   # the rejected blocks go back to the driver; `chosen` stays allocated
    return chosen, report


def _event_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def place_volume_stage_outputs(left, right, maxdisp, num_groups, max_tries=6, probe_reps=2, good_enough_tbs=6.3):
    """Output buffers ``(vol [B,G,D,H,W], disp [B,H,W], mean [B,D,H,W])`` for ``build_gwc_volume_with_disparity(..., out=...)``
    placed by timing that very launch on each candidate (``left`` / ``right`` are only read).  ``good_enough_tbs``: stop at the
    first candidate that streams the algorithmic bytes faster than this (TB/s): 6.3 is the fast level itself (4.64-4.72 ms at
    cfg-2); probe times taken while the other candidates are still held run a few percent above the steady state, so a looser
    threshold would stop at a middling arena."""
    from .cost_volume import build_gwc_volume_with_disparity
    B, C, H, W = left.shape
    D, G = int(maxdisp), int(num_groups)
    dev = left.device
    n_bytes = 4 * (B * G * D * H * W + B * D * H * W + B * H * W)
    algorithmic = n_bytes + 2 * B * C * H * W * left.element_size()

    def alloc():
        return (torch.empty((B, G, D, H, W), device=dev, dtype=torch.float32),
                torch.empty((B, H, W), device=dev, dtype=torch.float32),
                torch.empty((B, D, H, W), device=dev, dtype=torch.float32))

    def probe(bufs):
        with torch.no_grad():
            return _event_ms(lambda: build_gwc_volume_with_disparity(left, right, D, G, return_mean=True, out=bufs), probe_reps)

    bufs, report = place(alloc, probe, max_tries=max_tries, good_enough_ms=algorithmic / (good_enough_tbs * 1e12) * 1e3,
                         bytes_per_try=n_bytes, device=dev)
    report["policy"] = ("fastest of up to %d candidate allocations, each timed with the launch that will write it; stop at "
                        ">= %.1f TB/s; rejected blocks returned to the driver" % (max_tries, good_enough_tbs))
    report["bytes_per_candidate"] = n_bytes
    return bufs, report
