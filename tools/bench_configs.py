#!/usr/bin/env python3
"""Whole-chain pairs/s for the BASELINE.json configs that are NOT the headline line of bench.py (configs[2..4]), one JSON
line each, same fields as bench.py (N=1, synthetic inputs of the config's shapes, random-init weights, fp32):

  cfg-3  SceneFlow 960x540 D=192  CrossAttentionCostVolume (epipolar cross-attention + regression), batch 8
  cfg-4  SceneFlow 960x540 D=192  IterativeLocalCostVolume loop, 16 iterations, batch 8 per GPU (RAFT-style update
         operator at 1/4 resolution; and the LowCNN GRU updater at 1/8 resolution)
  cfg-5  KITTI 1242x375 -> 1248x384, D=256: cross-attention at 1/4 resolution (band 64) initialising the 16-iteration
         loop, batch 4 per GPU

The feature extractors are out of scope: the chains start from the tensors the extractors hand over.
`roofline` describes the dominant HAND-WRITTEN kernel of the chain, timed alone with HIP events; the loops themselves are
bounded by MIOpen's fp32 convolutions.   python tools/bench_configs.py [--steps K] > profiles/rNN_bench_configs.jsonl"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S  # noqa: E402

MFMA_F32_PEAK = 157.3  # TFLOP/s dense f32 MFMA (MI355X_MICROARCH.md)


def timed(fn, steps, warmup=2):
    with torch.no_grad():
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


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


def line(workload, pairs, sec, steps, roofline, extra=None):
    out = {"metric": "stereo pairs/s", "value": round(pairs / sec, 2), "unit": "pairs/s", "n_gpus": 1, "steps": steps,
           "warmup": 2, "ms_per_step": round(sec * 1e3, 3), "ms_per_pair": round(sec * 1e3 / pairs, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": workload, "pairs_per_gpu": pairs}, "roofline": roofline, "cpu_baseline": None}
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


def raft_inputs(B, Hq, Wq, dev):
    f1 = torch.randn(B, 256, Hq, Wq, device=dev)
    f2 = torch.randn(B, 256, Hq, Wq, device=dev)
    nets = [torch.tanh(torch.randn(B, 128, Hq >> i, Wq >> i, device=dev)) for i in range(3)]
    inps = [[torch.randn(B, 128, Hq >> i, Wq >> i, device=dev) for _ in range(3)] for i in range(3)]
    return f1, f2, nets, inps


def allpairs_roofline(f1, f2):
    B, C, Hq, Wq = f1.shape
    ms = event_ms(lambda: S.CorrBlock1D(f1, f2, 4, 4))
    tf = 2.0 * B * Hq * Wq * Wq * C / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "allpairs_lds_kernel + avgpool_pyramid_kernel (A8, once per pair batch)",
            "achieved": round(tf, 2), "peak": MFMA_F32_PEAK, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK, 4),
            "traffic": None, "avg_launch_ms": round(ms, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    want = lambda n: args.only is None or args.only in n

    if want("cfg3"):
        B, C, H, W, D = 8, 32, 540, 960, 192
        L = torch.randn(B, C, H, W, device=dev)
        R = torch.randn(B, C, H, W, device=dev)
        att = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")
        sec = timed(lambda: att(L, R), max(args.steps, 10))
        ms = event_ms(lambda: att(L, R))
        band = D * W - D * (D - 1) / 2
        tf = 2 * 2 * B * H * C * band / (ms * 1e-3) / 1e12
        line("cfg-3: SceneFlow 960x540 D=192 CrossAttentionCostVolume (QK^T / softmax / PV on f32 MFMA) + disparity "
             "regression, batch=8", B, sec, max(args.steps, 10),
             {"bound": "mfma", "kernel": "cross_attention_online_kernel<7,8> (A7)", "achieved": round(tf, 2),
              "peak": MFMA_F32_PEAK, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK, 4), "traffic": None,
              "avg_launch_ms": round(ms, 4), "algorithmic_flops": 2 * 2 * B * H * C * band})
        del L, R

    if want("cfg4"):
        B, Hq, Wq = 8, 136, 240  # 544x960 / 4
        f1, f2, nets, inps = raft_inputs(B, Hq, Wq, dev)
        roof = allpairs_roofline(f1, f2)
        for mp in (False, True):
            loop = S.RAFTStereoLoop([128, 128, 128], mixed_precision=mp, channels_last=True).eval().cuda()
            fn = lambda: loop(f1, f2, [n.clone() for n in nets], inps, iters=16, test_mode=True, use_graph=True)
            sec = timed(fn, args.steps)
            line("cfg-4: SceneFlow 960x540 (padded 544) IterativeLocalCostVolume loop (RAFT-style update operator at 1/4 "
                 "resolution, NHWC through the update block): all-pairs pyramid + 16 x (lookup + update block + coords) + convex "
                 "upsampling, batch=8 per GPU"
                 + (", update-block convolutions under fp16 autocast (the reference's mixed_precision flag; NOT the "
                    "parity configuration)" if mp else ", fp32"), B, sec, args.steps, roof,
                 {"bounded_by": "MIOpen convolutions of the update block (1.2 TFLOP per iteration for 8 pairs)",
                  "dtype": "f32 (+ f16 convs)" if mp else "f32"})
        del f1, f2, nets, inps
        vol = torch.randn(B, 24, 72, 120, device=dev)
        disp = torch.rand(B, 1, 72, 120, device=dev) * 20
        li = torch.randn(B, 3, 576, 960, device=dev)
        ri = torch.randn(B, 3, 576, 960, device=dev)
        lp = S.LowCNNGRULoop(24, 32, 20).eval().cuda()
        sec = timed(lambda: lp(vol, disp, li, ri, iters=16, use_graph=True), max(args.steps, 10))
        ms = event_ms(lambda: S.bounds_lookup(vol, disp, torch.rand(B, 2, 72, 120, device=dev) * 3, 20, True))
        line("cfg-4 (LowCNN GRU updater, baseline_dynamic_gru): 16 x DisparityUpdateDLCWithMask at 1/8 resolution "
             "[8,24,72,120] + convex upsampling x8 of the last iterate, batch=8 per GPU, fp32", B, sec, max(args.steps, 10),
             {"bound": "hbm", "kernel": "lookup_kernel<BOUNDS> (A11 sampler; latency-scale at this size)", "achieved": None,
              "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None, "avg_launch_ms": round(ms, 4)},
             {"bounded_by": "MIOpen convolutions (Winograd / GEMM) of the updater"})
        del vol, disp, li, ri

    if want("cfg5"):
        B, Hq, Wq, Dq = 4, 96, 312, 64  # 384x1248 / 4, D = 256 / 4
        f1, f2, nets, inps = raft_inputs(B, Hq, Wq, dev)
        att = S.CrossAttentionCostVolume(Dq, "neginf", "sqrt_c")
        loop = S.RAFTStereoLoop([128, 128, 128], channels_last=True).eval().cuda()

        def chain():
            _, d = att(f1, f2, need_out=False)
            init = torch.cat([-d.unsqueeze(1), torch.zeros_like(d.unsqueeze(1))], 1)  # raft_stereo.py:163-164
            return loop(f1, f2, [n.clone() for n in nets], inps, iters=16, flow_init=init, test_mode=True, use_graph=True)

        sec = timed(chain, args.steps)
        ms = event_ms(lambda: att(f1, f2, need_out=False))
        band = Dq * Wq - Dq * (Dq - 1) / 2
        tf = 2 * B * Hq * 256 * band / (ms * 1e-3) / 1e12
        line("cfg-5: KITTI 1242x375 (padded 1248x384) D=256: cross-attention at 1/4 resolution [4,256,96,312] (band 64) "
             "-> initial disparity -> 16-iteration IterativeLocalCostVolume loop, batch=4 per GPU, fp32", B, sec, args.steps,
             {"bound": "mfma", "kernel": "cross_attention_reg_kernel<3> (A7, C=256: HBM-leaning at 0.49 GB I/O)",
              "achieved": round(tf, 2), "peak": MFMA_F32_PEAK, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK, 4),
              "traffic": None, "avg_launch_ms": round(ms, 4)},
             {"bounded_by": "MIOpen convolutions of the update block"})


if __name__ == "__main__":
    main()
