#!/usr/bin/env python3
"""Whole-chain pairs/s for the BASELINE.json configs that are NOT the headline line of bench.py (configs[2..4]).

  cfg-3  SceneFlow 960x540 D=192  CrossAttentionCostVolume (epipolar cross-attention + regression), batch 8 per GPU
  cfg-4  SceneFlow 960x540 D=192  IterativeLocalCostVolume loop, 16 iterations, batch 8 per GPU -> 64 pairs over 8 GPUs
         (RAFT-style update operator at 1/4 resolution; and the LowCNN GRU updater at 1/8 resolution)
  cfg-5  KITTI 1242x375 -> 1248x384, D=256: cross-attention at 1/4 resolution (band 64) initialising the 16-iteration
         loop, batch 4 per GPU -> 32 pairs over 8 GPUs

`bench.py` calls `run_all()` after its headline workload and attaches the compact entries as `other_configs`; with N ranks
every rank runs its own shard of pairs (no data-path collective) and the time is the slowest rank's, exactly as for the
headline.  The feature extractors are out of scope: the chains start from the tensors the extractors hand over.
`roofline` describes the dominant HAND-WRITTEN kernel of the chain, timed alone with HIP events; the loops themselves are
bounded by MIOpen's fp32 convolutions.  `cpu_baseline` (rank 0, N=1): the C oracle, plus torch-CPU convolutions for the
update blocks, on a bounded sample scaled to one pair.

    python tools/bench_configs.py [--steps K] [--only cfg3|cfg4|cfg5] > profiles/rNN_bench_configs.jsonl
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S  # noqa: E402
from stereoformer_amd import shard  # noqa: E402
from stereoformer_amd import _native as NV  # noqa: E402

MFMA_F32_PEAK = 157.3  # TFLOP/s dense f32 MFMA (MI355X_MICROARCH.md)


def _sync(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def timed(fn, steps, warmup, world, device):
    """barrier + synchronize on both sides of exactly `steps` calls; returns (slowest rank's seconds per step, every
    rank's ms per step)."""
    with torch.no_grad():
        for _ in range(warmup):
            fn()
        _sync(world)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        _sync(world)
    sec = (time.perf_counter() - t0) / steps
    per_rank = shard.all_ranks(sec, device=device)
    return shard.max_over_ranks(sec, device=device), [round(x * 1e3, 3) for x in per_rank]


def event_ms(fn, reps=5):
    with torch.no_grad():
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def entry(name, workload, pairs_per_gpu, world, sec, per_rank, steps, warmup, roofline, cpu=None, extra=None):
    out = {"config": name, "workload": workload, "value": round(pairs_per_gpu * world / sec, 2), "unit": "pairs/s",
           "n_gpus": world, "pairs_per_gpu": pairs_per_gpu, "total_pairs_per_step": pairs_per_gpu * world,
           "steps": steps, "warmup": warmup, "ms_per_step": round(sec * 1e3, 3), "per_rank_ms_per_step": per_rank,
           "scaling": "weak", "dtype": "f32", "data": "synthetic", "roofline": roofline, "cpu_baseline": cpu}
    if extra:
        out.update(extra)
    return out


def pmc_traffic(fname, key_part):
    """HBM bytes per launch of a kernel from a COMMITTED counter pass under profiles/ (never measured in the bench run itself):
    (bytes, source) or (None, None)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", fname)) as f:
            d = json.load(f)
        for k, e in d.get("kernels", {}).items():
            if key_part in k and e.get("traffic_bytes_per_launch") is not None:
                return e["traffic_bytes_per_launch"], d.get("source")
    except (OSError, ValueError):
        pass
    return None, None


def raft_inputs(B, Hq, Wq, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    f1, f2 = rn(B, 256, Hq, Wq), rn(B, 256, Hq, Wq)
    nets = [torch.tanh(rn(B, 128, Hq >> i, Wq >> i)) for i in range(3)]
    inps = [[rn(B, 128, Hq >> i, Wq >> i) for _ in range(3)] for i in range(3)]
    return f1, f2, nets, inps


def allpairs_roofline(f1, f2):
    B, C, Hq, Wq = f1.shape
    ms = event_ms(lambda: S.CorrBlock1D(f1, f2, 4, 4))
    tf = 2.0 * B * Hq * Wq * Wq * C / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "allpairs_lds_kernel + avgpool_pyramid_kernel (A8, once per pair batch)",
            "achieved": round(tf, 2), "peak": MFMA_F32_PEAK, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK, 4),
            "traffic": None, "avg_launch_ms": round(ms, 4)}


# ----------------------------------------------------------------------------------------------------------- CPU legs
def _np(t):
    return t.detach().cpu().numpy()


@contextlib.contextmanager
def _cpu_ops():
    """BASELINE LEG ONLY: run the host modules (update blocks) on the CPU by routing their custom ops to the C oracle
    (numpy) -- convolutions / pooling stay torch-CPU.  Never active on the measured GPU path."""
    import oracle
    from stereoformer_amd import update_dlc, update_raft

    t = torch.from_numpy

    def gru_reset(h, rc, cr=None):
        return t(oracle.gru_reset(_np(h), _np(rc), None if cr is None else _np(cr)))

    def gru_blend(h, zc, qc, cz=None, cq=None, softsign=False):
        return t(oracle.gru_blend(_np(h), _np(zc), _np(qc), None if cz is None else _np(cz),
                                  None if cq is None else _np(cq), softsign))

    def resize(x, size, align_corners, scale=1.0):
        return t(oracle.resize_bilinear(_np(x), (int(size[0]), int(size[1])), bool(align_corners))) * scale

    def interp(x, dest):
        return resize(x, dest.shape[2:], True)

    def disp_warp(img, disp, padding_mode="border", check_positive=False):
        w, m = oracle.disp_warp(_np(img), _np(disp), padding_mode)
        return t(w), t(m)

    def uncertainty_volume(vol, disp):
        return t(oracle.uncertainty_volume(_np(vol), _np(disp)))

    def bounds_lookup(vol, disp, bounds, S_, consider_valid=True):
        b = _np(bounds)
        return t(oracle.bounds_lookup(_np(vol), _np(disp), b[:, :1], b[:, 1:], S_, consider_valid))

    patches = [(update_raft, "gru_reset", gru_reset), (update_raft, "gru_blend", gru_blend), (update_raft, "interp", interp),
               (update_dlc, "gru_reset", gru_reset), (update_dlc, "gru_blend", gru_blend),
               (update_dlc, "resize_bilinear", resize), (update_dlc, "disp_warp", disp_warp),
               (update_dlc, "uncertainty_volume", uncertainty_volume), (update_dlc, "bounds_lookup", bounds_lookup)]
    saved = [(m, n, getattr(m, n)) for m, n, _ in patches]
    try:
        for m, n, f in patches:
            setattr(m, n, f)
        yield oracle
    finally:
        for m, n, f in saved:
            setattr(m, n, f)


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_cross_attention(C, H, W, D, mask, scale, rows):
    """oracle cross-attention on `rows` rows of ONE pair, scaled to the H rows of a pair (rows are independent)."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(0)
    L = rng.standard_normal((1, C, rows, W), dtype=np.float32)
    R = rng.standard_normal((1, C, rows, W), dtype=np.float32)
    oracle.cross_attention(L[:, :, :2], R[:, :, :2], None, D, mask, scale)
    t0 = time.perf_counter()
    oracle.cross_attention(L, R, None, D, mask, scale)
    return (time.perf_counter() - t0) * H / rows


def cpu_raft_loop(Hq, Wq, iters, measured_iters=2, flow_init=None):
    """ONE pair: oracle all-pairs + pyramid, then `measured_iters` iterations of (oracle lookup + update block with torch-CPU
    convolutions and oracle gate math), scaled to `iters` iterations; + the convex upsampling once."""
    import numpy as np
    torch.manual_seed(0)
    f1, f2, nets, inps = raft_inputs(1, Hq, Wq, "cpu", 0)
    with _cpu_ops() as oracle, torch.no_grad():
        block = S.BasicMultiUpdateBlock_LZH(4, 4, 3, 2, [128, 128, 128]).eval()
        t0 = time.perf_counter()
        pyr = oracle.corr_pyramid(oracle.allpairs_corr(_np(f1), _np(f2)), 4)
        t_corr = time.perf_counter() - t0
        coords0 = np.stack(np.meshgrid(np.arange(Wq, dtype=np.float32), np.arange(Hq, dtype=np.float32)), 0)[None]
        coords1 = coords0.copy() if flow_init is None else coords0 + flow_init
        mask = None
        t0 = time.perf_counter()
        for _ in range(measured_iters):
            corr = torch.from_numpy(oracle.corr_lookup(pyr, coords1, 4, 4))
            flow = torch.from_numpy(coords1 - coords0)
            nets, mask, delta = block(nets, inps, corr, flow)
            d = _np(delta).copy()
            d[:, 1] = 0
            coords1 = coords1 + d
        t_iter = (time.perf_counter() - t0) / measured_iters
        t0 = time.perf_counter()
        oracle.upsample_convex(coords1 - coords0, _np(mask), 4)
        t_up = time.perf_counter() - t0
    return t_corr + iters * t_iter + t_up, {"allpairs_pyramid_s": round(t_corr, 3), "per_iteration_s": round(t_iter, 3),
                                            "upsample_s": round(t_up, 4)}


def cpu_lowcnn_loop(iters, measured_iters=2):
    torch.manual_seed(0)
    vol = torch.randn(1, 24, 72, 120)
    disp = torch.rand(1, 1, 72, 120) * 20
    li, ri = torch.randn(1, 3, 576, 960), torch.randn(1, 3, 576, 960)
    with _cpu_ops() as oracle, torch.no_grad():
        upd = S.DisparityUpdateDLCWithMask(24, 32, 20).eval()
        hidden = torch.zeros(1, 64, 72, 120)
        t0 = time.perf_counter()
        for _ in range(measured_iters):
            disp, hidden, mask = upd(vol, disp, li, ri, hidden, True)
        t_iter = (time.perf_counter() - t0) / measured_iters
        t0 = time.perf_counter()
        oracle.upsample_convex(_np(disp), _np(mask), 8)
        t_up = time.perf_counter() - t0
    return iters * t_iter + t_up, {"per_iteration_s": round(t_iter, 3), "upsample_s": round(t_up, 4)}


def _cpu_entry(seconds_per_pair, sample, detail=None):
    e = {"value": round(1.0 / seconds_per_pair, 4), "unit": "pairs/s", "cores": _cores(), "kind": "port",
         "sample": sample, "ms_per_pair": round(seconds_per_pair * 1e3, 1)}
    if detail:
        e["detail"] = detail
    return e


def _guard(fn):
    try:
        return fn()
    except Exception as e:  # a baseline leg must never take the measured entries down with it
        return {"error": f"{type(e).__name__}: {e}"}


def cpu_train_step_torch(C, H, W, D, G, rows):
    """BASELINE LEG ONLY ('port'): the cfg-2 training step on `rows` image rows of ONE pair in plain torch-CPU ops under autograd,
    following the reference's own formulations -- build_gwc_volume's per-disparity loop (GWC_Cost_Volume.py:5-24), mean over the
    groups, softmax expectation (estimation.py:7-35), LocalCostVolume's candidate sampling with straight-through ceil / floor
    (fixed_local_cost_volume.py:13-80) -- forward + backward w.r.t. both feature maps; seconds scaled to the full H rows."""
    torch.manual_seed(0)
    L = torch.randn(1, C, rows, W, requires_grad=True)
    R = torch.randn(1, C, rows, W, requires_grad=True)
    t0 = time.perf_counter()
    vol = L.new_zeros(1, G, D, rows, W)
    cpg = C // G
    for d in range(D):
        if d > 0:
            vol[:, :, d, :, d:] = (L[..., d:] * R[..., :-d]).view(1, G, cpg, rows, W - d).mean(2)
        else:
            vol[:, :, d] = (L * R).view(1, G, cpg, rows, W).mean(2)
    mean = vol.mean(1)
    disp = (torch.softmax(mean, 1) * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)).sum(1, keepdim=True)
    radius, S_ = 2.0, 20
    lower, upper = disp - radius, disp + radius
    invalid = ((lower < 0) | (upper >= D - 1)).float()
    cand = lower + (upper - lower) / S_ * torch.arange(S_ + 1, dtype=torch.float32).view(1, S_ + 1, 1, 1)
    cand = cand * (1 - invalid) + invalid * disp
    ce = (torch.ceil(cand) - cand).detach() + cand
    fl = (torch.floor(cand) - cand).detach() + cand
    ce, fl = ce.clamp(0, D - 1), fl.clamp(0, D - 1)
    fr = ce - cand
    val = torch.gather(mean, 1, fl.long()) * fr + torch.gather(mean, 1, ce.long()) * (1 - fr)
    out = (torch.softmax(val, 1) * cand).sum(1)
    torch.autograd.grad(out.mean(), (L, R))
    return (time.perf_counter() - t0) * H / rows


# ----------------------------------------------------------------------------------------------------------- configs
def cfg2_train(dev, rank, world, steps, warmup, cpu):
    """The cfg-2 chain as a TRAINING step (the reference's train_low.py:44-76 trains this head): features ->
    fused volume stage (group-wise volume + group mean + soft-argmin) -> LocalCostVolume -> scalar loss -> gradients w.r.t. both
    feature maps, 2 pairs per GPU (the group-wise volume and its transient gradient state are 6.4 GB per pair)."""
    B, C, H, W, D, G = 2, 32, 540, 960, 192, 8
    g = torch.Generator(device=dev).manual_seed(200 + rank)
    L = torch.randn(B, C, H, W, device=dev, generator=g)
    R = torch.roll(L, 9, 3) + 0.1 * torch.randn(B, C, H, W, device=dev, generator=g)
    lk = S.LocalCostVolume(2, 20)
    # the forward's output buffers live across the steps (a training loop owns them) and are placed like the headline's: the
    # 6.4-GB volume is the fastest window of a chunk-mapped arena; under grad `out=` writes them in place through history-free
    # aliases.  Any failure of the placement falls back to fresh allocations per step.
    bufs, placement = None, None
    try:
        with torch.no_grad():
            bufs, placement = S.hbm.place_volume_stage_outputs(L, R, D, G, mapped_span_bytes=96 << 30)
    except Exception as e:
        bufs, placement = None, {"mode": "none", "error": "%s: %s" % (type(e).__name__, e)}

    def step():
        with torch.enable_grad():
            Lt, Rt = L.detach().requires_grad_(True), R.detach().requires_grad_(True)
            _, dsp, mean = S.build_gwc_volume_with_disparity(Lt, Rt, D, G, return_mean=True, out=bufs)
            out = lk(mean, dsp.unsqueeze(1), True)
            return torch.autograd.grad(out.mean(), (Lt, Rt))

    sec, per_rank = timed(step, steps, warmup, world, dev)
    roof = None
    if rank == 0:
        with torch.no_grad():
            ms_f = event_ms(lambda: S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=bufs))
        vd = S.CostVolume(D)(L, R).detach().requires_grad_(True)
        with torch.enable_grad():
            dsp = S.DisparityEstimation(D)(vd)
        cd = torch.randn_like(dsp)

        def a4_bwd():
            with torch.enable_grad():
                torch.autograd.grad(dsp, vd, cd, retain_graph=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a4_bwd()
        e0.record()
        for _ in range(5):
            a4_bwd()
        e1.record()
        torch.cuda.synchronize()
        ms_b = e0.elapsed_time(e1) / 5
        n = B * H * W
        gb_f = (2 * n * C * 4 + n * G * D * 4 + n * D * 4 + n * 4) / 1e9
        gb_b = (2 * n * D * 4 + n * 4) / 1e9
        gl_, gr_ = torch.empty_like(L), torch.empty_like(R)
        gvol_ = torch.randn(B, D, H, W, device=dev)
        ms_a1 = event_ms(lambda: NV.call("stereo_gwc_volume_bwd_f32", NV.ptr(L), NV.ptr(R), NV.ptr(gvol_), NV.ptr(gl_), NV.ptr(gr_),
                                         B, C, H, W, D, 1, NV.stream_ptr(dev)))
        gb_a1 = (n * D * 4 + 4 * n * C * 4) / 1e9
        del gl_, gr_, gvol_
        roof = {"bound": "hbm", "kernel": "corr_volume_mfma_kernel (the fused volume stage, forward of the step: the group-wise "
                                          "volume is written once and never read back -- its backward runs through the mean volume)",
                "achieved": round(gb_f / ms_f * 1e3, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gb_f / ms_f * 1e3 / 8000.0, 4),
                "traffic": None, "avg_launch_ms": round(ms_f, 4), "algorithmic_bytes": int(gb_f * 1e9),
                "backward_kernels": {"soft_argmin_bwd_regs_kernel (A4 backward, one read + one write of the mean volume)":
                                     {"avg_launch_ms": round(ms_b, 4), "GB/s": round(gb_b / ms_b * 1e3, 1),
                                      "frac": round(gb_b / ms_b * 1e3 / 8000.0, 4)},
                                     "corr_volume_bwd_fused_kernel (A1 backward of the mean volume: dL and dR from one read of its "
                                     "gradient; bytes = that read + both features in + both gradients out)":
                                     {"avg_launch_ms": round(ms_a1, 4), "GB/s": round(gb_a1 / ms_a1 * 1e3, 1),
                                      "frac": round(gb_a1 / ms_a1 * 1e3 / 8000.0, 4)}}}
        del vd, dsp, cd
    cb = None
    if cpu and rank == 0:
        cb = _guard(lambda: _cpu_entry(cpu_train_step_torch(C, H, W, D, G, 40),
                                       "forward + backward of the step on 40 of the 540 rows of 1 pair (rows are independent), scaled "
                                       "x13.5; plain torch-CPU ops under autograd in the reference's own formulation"))
    return entry("cfg-2 (training step)", "SceneFlow 960x540 D=192: features -> group-wise volume + group mean + soft-argmin (one fused "
                 "launch) -> LocalCostVolume -> loss, forward + backward w.r.t. both feature maps, batch=2 per GPU, fp32", B, world,
                 sec, per_rank, steps, warmup, roof, cb, {"unit_note": "pairs/s = training pairs per second (forward + backward)",
                                                          "placement": placement})


def cfg3(dev, rank, world, steps, warmup, cpu):
    B, C, H, W, D = 8, 32, 540, 960, 192
    g = torch.Generator(device=dev).manual_seed(300 + rank)
    L = torch.randn(B, C, H, W, device=dev, generator=g)
    R = torch.randn(B, C, H, W, device=dev, generator=g)
    att = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")
    sec, per_rank = timed(lambda: att(L, R), steps, warmup, world, dev)
    roof = None
    if rank == 0:
        ms = event_ms(lambda: att(L, R))
        band = D * W - D * (D - 1) / 2
        flops = 2 * 2 * B * H * C * band
        tf = flops / (ms * 1e-3) / 1e12
        # the split-fp16 kernel executes each fp32 product as 3 fp16 piece products (hi hi | hi lo', lo' hi into two fp32
        # accumulators): 3x the algorithmic flops on the fp16 pipe.  `frac` is against the fp32 matrix peak (the yardstick for an
        # fp32-accurate result); the executed piece-product rate against the fp16 peak is given beside it.  The kernel is bound
        # by its VECTOR work (exp2, the splits of P, the online-softmax bookkeeping: ~135 VALU instructions per 12 MFMAs), not by
        # the matrix pipe -- halving the MFMAs (22 -> 12 per key tile against round 2's split-bf16 form) moved it by 5 %.
        roof = {"bound": "mfma", "kernel": "cross_attention_split16_kernel<7,8,float> (A7 on v_mfma_f32_32x32x16_f16, hi + 2^-11 lo' operands)",
                "achieved": round(tf, 2), "peak": MFMA_F32_PEAK, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK, 4),
                "traffic": pmc_traffic("pmc_a7_latest.json", "cfg3: ")[0], "traffic_source": pmc_traffic("pmc_a7_latest.json", "cfg3: ")[1],
                "algorithmic_bytes": 3 * B * C * H * W * 4 + B * H * W * 4,
                "avg_launch_ms": round(ms, 4), "algorithmic_flops": flops,
                "executed_f16_tflops": round(3 * tf, 1), "frac_of_f16_peak_2500": round(3 * tf / 2500.0, 4)}
    del L, R
    cb = None
    if cpu and rank == 0:
        cb = _guard(lambda: _cpu_entry(cpu_cross_attention(C, H, W, D, "zero", "sqrt_c", 90),
                                       "90 of the 540 rows of 1 pair (rows are independent), scaled x6; C oracle, OpenMP"))
    return entry("cfg-3", "SceneFlow 960x540 D=192 CrossAttentionCostVolume (QK^T / softmax / PV on the matrix cores, fp32-accurate "
                 "split-fp16 form) + disparity regression, batch=8 per GPU", B, world, sec, per_rank, steps, warmup, roof, cb)


def conv_split_roofline(B, Hq, Wq, dev):
    """The launch that dominates a RAFT-loop iteration: convz | convr of the 1/4-scale GRU over [h | motion | interp] (384 -> 256
    channels, 3x3, gate math in the epilogue) on the split-fp16 kernel.  `achieved` counts fp32-equivalent flops; the matrix
    cores execute 3 fp16 piece products per fp32 product, so the dense fp16 peak / 3 is the peak of this formulation."""
    import torch.nn as nn
    from stereoformer_amd import conv_split as C
    cl = torch.channels_last
    srcs = [torch.randn(B, 128, Hq, Wq, device=dev).contiguous(memory_format=cl) for _ in range(3)]
    ctx = [torch.randn(B, 128, Hq, Wq, device=dev).contiguous(memory_format=cl) for _ in range(2)]
    convs = [nn.Conv2d(384, 128, 3, padding=1).to(dev) for _ in range(2)]
    epi = [("sigmoid", ctx[0]), ("sigmoid_mul", ctx[1], srcs[0])]
    with torch.no_grad():
        ms = event_ms(lambda: C.conv_nhwc(srcs, convs, epilogue=epi))
        ms6 = event_ms(lambda: C.conv_nhwc(srcs, convs, epilogue=epi, precision="bf16x6"))
    gf = 2 * B * Hq * Wq * 384 * 256 * 9 / 1e9
    tf = gf / ms
    traffic = None
    try:   # HBM bytes per launch from the committed counter passes of this kernel (tools/refresh_profiles.sh), not this run
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_conv_latest.json")) as f:
            traffic = json.load(f)
    except (OSError, ValueError):
        pass
    return {"bound": "mfma", "kernel": "conv_split_kernel<3,4,F32_F16X3> (A14: convz | convr of the 1/4-scale ConvGRU, 384 -> 256 "
                                       "channels, sigmoid / sigmoid x h epilogues; fp32-faithful split-fp16 on "
                                       "v_mfma_f32_32x32x16_f16, 3 piece products; most of the loop's time is this kernel)",
            "achieved": round(tf, 1), "peak": round(2500.0 / 3, 1), "unit": "TFLOP/s", "frac": round(tf / (2500.0 / 3), 4),
            "traffic": None if traffic is None else traffic.get("traffic_bytes_per_launch"),
            "traffic_source": None if traffic is None else traffic.get("source"),
            "algorithmic_bytes": B * Hq * Wq * 128 * 4 * (3 + 3 + 2) + 384 * 256 * 9 * 4,
            "avg_launch_ms": round(ms, 4), "executed_f16_tflops": round(3 * tf, 1),
            "frac_of_f32_mfma_peak": round(tf / MFMA_F32_PEAK, 4),
            "split_bf16_six_product_form_ms": round(ms6, 4),
            "sustained_mfma_tflops_random_operands": [1470, 1660],
            "frac_of_sustained_rate": round(3 * tf / 1660.0, 3),
            "note": "fp32-equivalent flops; MIOpen's fp32 implicit GEMM runs the same convolution at ~128 TFLOP/s (0.81 of the f32 "
                    "MFMA peak), round 2's six-product split-bf16 form at ~220.  The 2.5 PFLOP/s figure is not reachable with real "
                    "data on this part: a pure 32x32x16 MFMA loop sustains 2.43 PFLOP/s on CONSTANT operands but 1.47-1.66 PFLOP/s "
                    "on random bits (power: the clock drops to 1.4-1.6 GHz; tools/mfma_rate_bf16.hip) -- frac_of_sustained_rate is "
                    "against 1.66"}


def cfg4(dev, rank, world, steps, warmup, cpu, mixed=False):
    B, Hq, Wq = 8, 136, 240  # 544x960 / 4
    f1, f2, nets, inps = raft_inputs(B, Hq, Wq, dev, 400 + rank)
    roof = conv_split_roofline(B, Hq, Wq, dev) if rank == 0 else None
    torch.manual_seed(4)
    out = []
    for mp in ((False, True, "high") if mixed else (False,)):
        hp = mp == "high"     # fp32 activations, bf16x3 convolution arithmetic (conv_precision='high')
        mp = mp is True
        loop = S.RAFTStereoLoop([128, 128, 128], mixed_precision=mp,      # (defaults: NHWC + conv_split under no_grad)
                                conv_precision="high" if hp else "highest").eval().to(dev)
        fn = lambda: loop(f1, f2, [n.clone() for n in nets], inps, iters=16, test_mode=True, use_graph=True)
        sec, per_rank = timed(fn, steps, warmup, world, dev)
        cb = None
        if cpu and rank == 0 and not mp and not hp:
            def leg():
                s, d = cpu_raft_loop(Hq, Wq, 16)
                return _cpu_entry(s, "1 pair: all-pairs + pyramid once, 2 measured iterations scaled to 16, upsampling once; C oracle "
                                     "+ torch-CPU convolutions", d)
            cb = _guard(leg)
        out.append(entry("cfg-4" + (" (fp16 autocast convs)" if mp else (" (bf16x3 convs)" if hp else "")),
                         "SceneFlow 960x540 (padded 544) IterativeLocalCostVolume loop (RAFT-style update operator at 1/4 "
                         "resolution, NHWC through the update block): all-pairs pyramid + 16 x (lookup + update block + coords) + "
                         "convex upsampling, batch=8 per GPU (64 pairs over 8 GPUs)"
                         + (", update-block convolutions under fp16 autocast (the reference's mixed_precision flag; NOT the "
                            "parity configuration)" if mp else
                            (", fp32 activations with conv_precision='high': two bf16 pieces / three products per fp32 product "
                             "(~2^-16 relative per product; NOT the parity configuration)" if hp else ", fp32")),
                         B, world, sec, per_rank, steps, warmup, roof, cb,
                         {"bounded_by": ("MIOpen fp16 convolutions of the update block" if mp else
                                         "the update block's convolutions (1.2 TFLOP fp32-equivalent per iteration for 8 pairs) on "
                                         "the split-fp16 matrix-core kernel"),
                          "dtype": "f32 (+ f16 convs)" if mp else "f32"}))
        del loop
    del f1, f2, nets, inps
    g = torch.Generator(device=dev).manual_seed(450 + rank)
    vol = torch.randn(B, 24, 72, 120, device=dev, generator=g)
    disp = torch.rand(B, 1, 72, 120, device=dev, generator=g) * 20
    li = torch.randn(B, 3, 576, 960, device=dev, generator=g)
    ri = torch.randn(B, 3, 576, 960, device=dev, generator=g)
    lp = S.LowCNNGRULoop(24, 32, 20).eval().to(dev)
    n = max(steps, 20)
    sec, per_rank = timed(lambda: lp(vol, disp, li, ri, iters=16, use_graph=True), n, max(warmup, 5), world, dev)
    roof2 = None
    if rank == 0:
        # the loop's dominant hand-written launch: conv_z | conv_b of the updater's ConvGRU over [err feat | unc feat | h]
        # (32 + 32 + 64 -> 64 + 64 channels, gate epilogues) at [8,.,72,120] -- 4 of the 9 launches of an iteration are this kernel
        import torch.nn as nn
        from stereoformer_amd import conv_split as C
        cl = torch.channels_last
        srcs = [torch.randn(B, c, 72, 120, device=dev).contiguous(memory_format=cl) for c in (32, 32, 64)]
        convs = [nn.Conv2d(128, 64, 3, padding=1).to(dev) for _ in range(2)]
        epi = [("sigmoid", None), ("sigmoid_mul", None, srcs[2])]
        with torch.no_grad():
            ms = event_ms(lambda: C.conv_nhwc(srcs, convs, epilogue=epi), reps=20)
        gf = 2 * B * 72 * 120 * 128 * 128 * 9 / 1e9
        tf = gf / ms
        roof2 = {"bound": "mfma", "kernel": "conv_split_kernel<3,2,F32_F16X3> (A11: conv_z | conv_b of the updater's ConvGRU, 128 -> 64 + 64 "
                                            "channels at [8,.,72,120], sigmoid / sigmoid x h epilogues; 2-row pixel tiles)",
                 "achieved": round(tf, 1), "peak": round(2500.0 / 3, 1), "unit": "TFLOP/s", "frac": round(tf / (2500.0 / 3), 4),
                 "traffic": None, "avg_launch_ms": round(ms, 4),
                 "algorithmic_bytes": B * 72 * 120 * 4 * (128 + 64 + 128) + 128 * 128 * 9 * 4,
                 "note": "a 20-GFLOP launch of 1 152 workgroups: tile quantisation and the 8-chunk K loop's prologue / epilogue, not "
                         "the matrix pipe, set its time; per iteration the loop is 9 launches (guidance inputs, two encoder "
                         "convolutions, two GRU launches, offset conv1 / conv2, the bounds sampler)"}
        del srcs, convs
    cb = None
    if cpu and rank == 0:
        def leg2():
            s, d = cpu_lowcnn_loop(16)
            return _cpu_entry(s, "1 pair: 2 measured iterations scaled to 16, upsampling once; C oracle + torch-CPU convolutions", d)
        cb = _guard(leg2)
    out.append(entry("cfg-4 (LowCNN GRU updater)", "16 x DisparityUpdateDLCWithMask (baseline_dynamic_gru) at 1/8 resolution "
                     "[8,24,72,120] + convex upsampling x8 of the last iterate, batch=8 per GPU, fp32", B, world, sec, per_rank, n,
                     warmup, roof2, cb, {"bounded_by": "the updater's convolutions (3x3: csrc/conv_split.hip; encoder and 1x1 mask projection: MIOpen) "
                                      "and ~30 small launches per iteration"}))
    return out


def cfg5(dev, rank, world, steps, warmup, cpu):
    B, Hq, Wq, Dq = 4, 96, 312, 64  # 384x1248 / 4, D = 256 / 4
    f1, f2, nets, inps = raft_inputs(B, Hq, Wq, dev, 500 + rank)
    att = S.CrossAttentionCostVolume(Dq, "neginf", "sqrt_c")
    torch.manual_seed(5)
    loop = S.RAFTStereoLoop([128, 128, 128]).eval().to(dev)

    def chain():
        _, d = att(f1, f2, need_out=False)
        init = torch.cat([-d.unsqueeze(1), torch.zeros_like(d.unsqueeze(1))], 1)  # raft_stereo.py:163-164
        return loop(f1, f2, [n.clone() for n in nets], inps, iters=16, flow_init=init, test_mode=True, use_graph=True)

    sec, per_rank = timed(chain, steps, warmup, world, dev)
    roof = None
    if rank == 0:
        ms = event_ms(lambda: att(f1, f2, need_out=False))
        band = Dq * Wq - Dq * (Dq - 1) / 2
        tf = 2 * B * Hq * 256 * band / (ms * 1e-3) / 1e12
        gbs = (2 * B * 256 * Hq * Wq * 4 + B * Hq * Wq * 4) / (ms * 1e-3) / 1e9
        # 11 flop/B at C = 256: left of the f32 ridge, so HBM is the roofline; the matrix-core figure is reported beside it
        roof = {"bound": "hbm", "kernel": "cross_attention_chunked_split_kernel<3,*> at C=256 (A7, disparity only: the QK^T half; "
                                          "split-bf16 operands on v_mfma_f32_32x32x16_bf16)",
                "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                "traffic": pmc_traffic("pmc_a7_latest.json", "cfg5: ")[0], "traffic_source": pmc_traffic("pmc_a7_latest.json", "cfg5: ")[1],
                "algorithmic_bytes": 2 * B * 256 * Hq * Wq * 4 + B * Hq * Wq * 4,
                "avg_launch_ms": round(ms, 4), "fp32_equivalent_tflops": round(tf, 2),
                "frac_of_f32_mfma_peak": round(tf / MFMA_F32_PEAK, 4)}
    cb = None
    if cpu and rank == 0:
        def leg():
            a = cpu_cross_attention(256, Hq, Wq, Dq, "neginf", "sqrt_c", 24)
            s, d = cpu_raft_loop(Hq, Wq, 16)
            d["cross_attention_s"] = round(a, 3)
            return _cpu_entry(a + s, "1 pair: cross-attention on 24 of 96 rows scaled x4 + all-pairs once + 2 measured loop iterations "
                                     "scaled to 16; C oracle + torch-CPU convolutions", d)
        cb = _guard(leg)
    return entry("cfg-5", "KITTI 1242x375 (padded 1248x384) D=256: cross-attention at 1/4 resolution [4,256,96,312] (band 64) "
                 "-> initial disparity -> 16-iteration IterativeLocalCostVolume loop, batch=4 per GPU (32 pairs over 8 GPUs), fp32",
                 B, world, sec, per_rank, steps, warmup, roof, cb,
                 {"bounded_by": "the update block's convolutions (split-fp16 matrix-core kernel, csrc/conv_split.hip)"})


def run_all(dev, rank=0, world=1, cpu=False, rehearsal=False, steps=3, warmup=2, only=None, mixed=False):
    """Compact entries for cfg-3/4/5.  A failing config is reported in place, it never raises into the headline line."""
    want = lambda n: only is None or only in n
    out = []
    # the short steps (0.7 - 6 ms) get enough warm-up and timed steps to sit past the clock ramp of an idle GPU (the first ~20
    # calls of the 0.74-ms attention step run at 0.95-0.8 ms: tools/a7_sustained.py); the 30 - 65-ms loops need no such care
    for name, fn in (("cfg2train", lambda: [cfg2_train(dev, rank, world, max(steps, 40), max(warmup, 15), cpu)]),
                     ("cfg3", lambda: [cfg3(dev, rank, world, max(steps, 100), max(warmup, 30), cpu)]),
                     ("cfg4", lambda: cfg4(dev, rank, world, steps, warmup, cpu, mixed)),
                     ("cfg5", lambda: [cfg5(dev, rank, world, steps, warmup, cpu)])):
        if not want(name):
            continue
        try:
            res = fn()
        except Exception as e:
            res = [{"config": name, "error": f"{type(e).__name__}: {e}"}]
        if rehearsal:
            for r in res:
                r["data"] = "synthetic (REHEARSAL: all ranks share cuda:0; not a measurement)"
        out.extend(res)
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--only", default=None)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for e in run_all(dev, 0, 1, cpu=args.cpu, steps=args.steps, only=args.only, mixed=True):
        print(json.dumps(e), flush=True)


if __name__ == "__main__":
    main()
