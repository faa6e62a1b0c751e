#!/usr/bin/env python3
"""Per-kernel timing at the BASELINE config sizes, with algorithmic bytes/flops (SURVEY 8(d),
DESIGN.md section 4) and the fraction of the bounding roofline.  Writes a table to stdout and
JSON to gpurun_out/kernel_bench.json.  Usage: python tools/kernel_bench.py [--only name]"""
import argparse
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch

import stereoformer_amd as S

HBM = 8000.0      # GB/s spec
MFMA_F32 = 157.3  # TFLOP/s (f32-in MFMA == f32 VALU peak)


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def timeit_graph(fn, n=20, replays=10):
    """For launches of < 0.1 ms: n calls captured in a graph and replayed (the Python loop of `timeit` cannot feed the queue that
    fast and would report the host).  Not under rocprofv3 (graph capture under the profiler hung): --no-graph there."""
    if NO_GRAPH:
        return timeit(fn, 50, 10)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * replays)


NO_GRAPH = False


def main():
    global NO_GRAPH
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-graph", action="store_true", help="time small launches in a host loop (when run under rocprofv3)")
    ap.add_argument("--only", default=None)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    NO_GRAPH = args.no_graph
    dev = "cuda"
    rows = []

    def add(name, ms, gbytes=None, gflop=None, bound="hbm", note=""):
        r = {"kernel": name, "ms": round(ms, 4), "bound": bound, "note": note}
        if gbytes is not None:
            r["GB"] = round(gbytes, 4)
            r["GB/s"] = round(gbytes / ms * 1e3, 1)
        if gflop is not None:
            r["GFLOP"] = round(gflop, 2)
            r["TFLOP/s"] = round(gflop / ms, 2)
        if bound == "hbm" and gbytes is not None:
            r["frac"] = round(gbytes / ms * 1e3 / HBM, 4)
        if bound == "mfma" and gflop is not None:
            r["frac"] = round(gflop / ms / MFMA_F32, 4)
        rows.append(r)
        print(json.dumps(r), flush=True)

    def want(n):
        return args.only is None or args.only in n

    B, C, H, W, D, G = 8, 32, 540, 960, 192, 8
    N = B * H * W
    torch.manual_seed(0)
    L = torch.randn(B, C, H, W, device=dev)
    R = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        if want("A1_corr_volume"):
            add("A1_corr_volume cfg-2", timeit(lambda: S.CostVolume(D)(L, R), args.reps),
                (2 * N * C * 4 + N * D * 4) / 1e9, 2 * B * H * C * (D * W - D * (D - 1) / 2) / 1e9)
        if want("A1_corr_volume"):   # the reference's own use: 256 channels at 1/8 scale, D = 192 // 8 (baseline.py:63,114)
            Bn, Cn, Hn, Wn, Dn = 8, 256, 68, 120, 24
            Ln, Rn = torch.randn(Bn, Cn, Hn, Wn, device=dev), torch.randn(Bn, Cn, Hn, Wn, device=dev)
            nb = (2 * Bn * Cn * Hn * Wn * 4 + Bn * Dn * Hn * Wn * 4) / 1e9
            nf = 2 * Bn * Hn * Cn * (Dn * Wn - Dn * (Dn - 1) / 2) / 1e9
            add("A1_corr_volume reference-native [8,256,68,120] D=24", timeit(lambda: S.CostVolume(Dn)(Ln, Rn), 50, 5), nb, nf,
                note="channel-chunked matrix-core kernel")
            add("A1+A4 corr_volume_with_disparity reference-native", timeit(lambda: S.corr_volume_with_disparity(Ln, Rn, Dn), 50, 5),
                nb + Bn * Hn * Wn * 4 / 1e9, nf)
            with S._native.test_options(corr_mfma=0):
                add("A1_corr_volume reference-native [VALU kernel, for comparison]", timeit(lambda: S.CostVolume(Dn)(Ln, Rn), 50, 5), nb, nf)
            Lh, Rh = Ln.half(), Rn.half()
            add("A1_corr_volume reference-native, fp16 features", timeit(lambda: S.CostVolume(Dn)(Lh, Rh), 50, 5),
                (2 * Bn * Cn * Hn * Wn * 2 + Bn * Dn * Hn * Wn * 4) / 1e9, nf)
            del Ln, Rn, Lh, Rh
        if want("A3_gwc_volume"):
            add("A3_gwc_volume cfg-2 G=8", timeit(lambda: S.build_gwc_volume(L, R, D, G), 5),
                (2 * N * C * 4 + N * G * D * 4) / 1e9)
        vol = S.CostVolume(D)(L, R)
        if want("A4_soft_argmin"):
            add("A4_soft_argmin cfg-2", timeit(lambda: S.DisparityEstimation(D)(vol), args.reps), (N * D * 4 + N * 4) / 1e9)
            add("A4_soft_argmin_with_prob cfg-2", timeit(lambda: S.DisparityEstimationWithProb(D)(vol), args.reps),
                (2 * N * D * 4 + N * 4) / 1e9, note="single pass: softmax column held in registers (1 read + 1 write)")
        disp = S.DisparityEstimation(D)(vol).unsqueeze(1)
        if want("A5_local_lookup"):
            add("A5_local_lookup r=2 S=20 cfg-2", timeit(lambda: S.LocalCostVolume(2, 20)(vol, disp, True), args.reps),
                (N * D * 4 + 2 * N * 4) / 1e9, note="upper-bound bytes (whole volume); gather-only bound 0.73 GB")
        if want("A6_variance_lookup"):
            add("A6_variance_lookup g=1 S=10 cfg-2",
                timeit(lambda: S.Variance_Based_Local_Cost_Volume(1, 10)(vol, disp, True), args.reps),
                (N * D * 4 + 2 * N * 4) / 1e9)
        if want("A4_group_mean"):
            volg = S.build_gwc_volume(L, R, D, G)
            add("group_mean_soft_argmin cfg-2", timeit(lambda: S.group_mean_soft_argmin(volg, return_mean=True), 5),
                (N * G * D * 4 + N * D * 4 + N * 4) / 1e9)
            add("group_mean_soft_argmin(no mean out) cfg-2", timeit(lambda: S.group_mean_soft_argmin(volg), 5),
                (N * G * D * 4 + N * 4) / 1e9)
            del volg
        if want("A1A4_fused"):
            add("A1+A4 fused corr_soft_argmin cfg-2", timeit(lambda: S.corr_soft_argmin(L, R, D), args.reps),
                (2 * N * C * 4 + N * 4) / 1e9, 2 * B * H * C * (D * W - D * (D - 1) / 2) / 1e9, bound="mfma",
                note="= cross-attention v5 kernel, mask zero, scale 1/C, no output tensor (identity I3)")
        if want("A2_"):
            L1, R1 = L[:1, :8].contiguous(), R[:1, :8].contiguous()
            n1 = H * W
            add("A2_diff_volume B1 C8 D=192", timeit(lambda: S.CostVolume(D, "difference")(L1, R1), 5),
                (2 * n1 * 8 * 4 + n1 * 8 * D * 4) / 1e9)
            add("A2_concat_volume B1 C8 D=192", timeit(lambda: S.CostVolume(D, "concat")(L1, R1), 5),
                (2 * n1 * 8 * 4 + n1 * 16 * D * 4) / 1e9)
        del vol
        if want("A7_cross_attention"):
            att = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")
            band = D * W - D * (D - 1) / 2
            add("A7_cross_attention cfg-3 C=32 D=192", timeit(lambda: att(L, R), 10, 2),
                (4 * N * C * 4 + N * 4) / 1e9, 2 * 2 * B * H * C * band / 1e9, bound="mfma",
                note="split-bf16 kernel (v7); frac = fp32-equivalent useful flops / f32 MFMA peak")
            add("A7_cross_attention(disp only) cfg-3", timeit(lambda: att(L, R, need_out=False), 10, 2),
                (2 * N * C * 4 + N * 4) / 1e9, 2 * B * H * C * band / 1e9, bound="mfma")
            with S._native.test_options(att_kernel=5):
                add("A7_cross_attention cfg-3 [fp32 online-softmax kernel v5, for comparison]", timeit(lambda: att(L, R), 10, 2),
                    (4 * N * C * 4 + N * 4) / 1e9, 2 * 2 * B * H * C * band / 1e9, bound="mfma")
        # ---- RAFT shapes (cfg-4 per GPU)
        Br, Cr, Hr, Wr = 8, 256, 136, 240
        f1 = torch.randn(Br, Cr, Hr, Wr, device=dev)
        f2 = torch.randn(Br, Cr, Hr, Wr, device=dev)
        if want("A7_cross_attention"):
            # cfg-3 secondary shape (RAFT-fnet-like features, quarter resolution) and cfg-5 (KITTI 384x1248 / 4, D/4 = 64)
            for (nm, bb, cc, hh, ww, dd) in (("cfg-3b [8,256,136,240] D=48", 8, 256, 136, 240, 48),
                                             ("cfg-5 [4,256,96,312] D=64", 4, 256, 96, 312, 64)):
                q1 = torch.randn(bb, cc, hh, ww, device=dev)
                q2 = torch.randn(bb, cc, hh, ww, device=dev)
                a2 = S.CrossAttentionCostVolume(dd, "neginf", "sqrt_c")
                bnd = dd * ww - dd * (dd - 1) / 2
                gb, gf = (4 * bb * cc * hh * ww * 4 + bb * hh * ww * 4) / 1e9, 2 * 2 * bb * hh * cc * bnd / 1e9
                ms = timeit(lambda: a2(q1, q2), 20, 3)
                # 11 flop/B: left of the f32 ridge (~20 flop/B) -> the HBM roofline is the relevant one; both are reported
                add(f"A7_cross_attention {nm}", ms, gb, gf, bound="hbm",
                    note=f"channel-chunked item pipeline, split-bf16 form (v8); bytes = L + R + V + out; "
                         f"{gf / ms / MFMA_F32:.3f} of the f32 MFMA peak on fp32-equivalent algorithmic flops")
                with torch.no_grad():
                    ms_d = timeit_graph(lambda: a2(q1, q2, need_out=False))
                add(f"A7_cross_attention(disp only) {nm}", ms_d,
                    (2 * bb * cc * hh * ww * 4 + bb * hh * ww * 4) / 1e9, gf / 2, bound="hbm",
                    note="attention_chunked16.hip: split-fp16 operands, 16-channel items, one accumulator; timed as a replayed "
                         "graph of 20 launches; bytes = L + R + disparity")
                with S._native.test_options(att_kernel=8), torch.no_grad():
                    add(f"A7_cross_attention(disp only) {nm} [split-bf16 chunked kernel, round 2's, for comparison]",
                        timeit_graph(lambda: a2(q1, q2, need_out=False)),
                        (2 * bb * cc * hh * ww * 4 + bb * hh * ww * 4) / 1e9, gf / 2, bound="hbm")
                with S._native.test_options(att_kernel=4):
                    add(f"A7_cross_attention {nm} [two-pass fp32 kernel v4, for comparison]", timeit(lambda: a2(q1, q2), 20, 3),
                        gb, gf, bound="hbm")
                with S._native.test_options(att_kernel=6):
                    add(f"A7_cross_attention {nm} [chunked pipeline, fp32-MFMA form v6, for comparison]",
                        timeit(lambda: a2(q1, q2), 20, 3), gb, gf, bound="hbm")
                del q1, q2
        if want("N2_fp16"):   # fp16 feature operands (fp32 arithmetic / outputs) + the extractor epilogue
            Lh, Rh = L.half(), R.half()
            add("N2 A1_corr_volume cfg-2, fp16 features", timeit(lambda: S.CostVolume(D)(Lh, Rh), args.reps),
                (2 * N * C * 2 + N * D * 4) / 1e9)
            add("N2 A3+mean+A4 fused cfg-2, fp16 features", timeit(lambda: S.build_gwc_volume_with_disparity(Lh, Rh, D, G, return_mean=True), 5),
                (2 * N * C * 2 + N * G * D * 4 + N * D * 4 + N * 4) / 1e9)
            add("N2 A1+A4 corr_soft_argmin cfg-2, fp16 features", timeit(lambda: S.corr_soft_argmin(Lh, Rh, D), args.reps),
                (2 * N * C * 2 + N * 4) / 1e9, 2 * B * H * C * (D * W - D * (D - 1) / 2) / 1e9, bound="mfma")
            att16 = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")
            add("N2 A7_cross_attention cfg-3, fp16 features", timeit(lambda: att16(Lh, Rh), 5, 1),
                (2 * N * C * 2 + N * C * 4 + N * 4) / 1e9, 2 * 2 * B * H * C * (D * W - D * (D - 1) / 2) / 1e9, bound="mfma")
            raw = torch.randn(B, C, H, W, device=dev)
            add("N2 instance_norm_relu [8,32,540,960] -> fp16", timeit(lambda: S.instance_norm_relu(raw), args.reps),
                (N * C * (4 + 4 + 2)) / 1e9, note="two reads of x (statistics, apply) + one fp16 write")
            add("N2 instance_norm_relu [8,32,540,960] -> fp32", timeit(lambda: S.instance_norm_relu(raw, out_dtype=torch.float32), args.reps),
                (N * C * (4 + 4 + 4)) / 1e9)
            rawcl = raw.contiguous(memory_format=torch.channels_last)
            add("N2 instance_norm_relu [8,32,540,960] channels-last in -> NCHW fp16", timeit(lambda: S.instance_norm_relu(rawcl), args.reps),
                (N * C * (4 + 4 + 2)) / 1e9, note="the layout change rides on the second read (no transpose pass)")
            add("N2 [for comparison] .contiguous() of the channels-last tensor alone", timeit(lambda: rawcl.contiguous(), args.reps),
                (N * C * (4 + 4)) / 1e9)
            add("N2 relayout channels-last -> NCHW [8,32,540,960] fp32 (N.nchw)", timeit(lambda: S._native.nchw(rawcl), args.reps),
                (N * C * (4 + 4)) / 1e9)
            del rawcl
            del Lh, Rh, raw
            f1h, f2h = f1.half(), f2.half()
            pyr_b = Br * Hr * Wr * (Wr + Wr // 2 + Wr // 4 + Wr // 8 + Wr // 16) * 4
            add("N2 A8_allpairs+pyramid cfg-4, fp16 features", timeit(lambda: S.CorrBlock1D(f1h, f2h, 4, 4), args.reps),
                (2 * Br * Cr * Hr * Wr * 2 + pyr_b) / 1e9, 2 * Br * Hr * Wr * Wr * Cr / 1e9, bound="mfma")
            a16 = S.CrossAttentionCostVolume(48, "neginf", "sqrt_c")
            bnd = 48 * Wr - 48 * 47 / 2
            add("N2 A7_cross_attention cfg-3b [8,256,136,240] D=48, fp16 features", timeit(lambda: a16(f1h, f2h), 20, 3),
                (3 * Br * Cr * Hr * Wr * 2 + Br * Cr * Hr * Wr * 4 + Br * Hr * Wr * 4) / 1e9, 2 * 2 * Br * Hr * Cr * bnd / 1e9)
            del f1h, f2h
        if want("A8_allpairs"):
            pyr_bytes = Br * Hr * Wr * (Wr + Wr // 2 + Wr // 4 + Wr // 8 + Wr // 16) * 4
            add("A8_allpairs+pyramid cfg-4 [8,256,136,240]", timeit(lambda: S.CorrBlock1D(f1, f2, 4, 4), args.reps),
                (2 * Br * Cr * Hr * Wr * 4 + pyr_bytes) / 1e9, 2 * Br * Hr * Wr * Wr * Cr / 1e9, bound="mfma")
        if want("A8_allpairs"):
            lv0 = torch.empty(Br * Hr * Wr, 1, 1, Wr, device=dev)
            add("A8_allpairs level 0 only cfg-4", timeit(lambda: S.CorrBlock1D.corr(f1, f2), args.reps),
                (2 * Br * Cr * Hr * Wr * 4 + Br * Hr * Wr * Wr * 4) / 1e9, 2 * Br * Hr * Wr * Wr * Cr / 1e9, bound="mfma")
            del lv0
        blk = S.CorrBlock1D(f1, f2, 4, 4)
        coords = S.coords_grid(Br, Hr, Wr, device=dev) - torch.rand(Br, 2, Hr, Wr, device=dev) * 30
        coords = coords.contiguous()
        if want("A9_corr_lookup"):
            add("A9_corr_lookup cfg-4", timeit(lambda: blk(coords), args.reps),
                (Br * 36 * Hr * Wr * 4 * 3) / 1e9, note="out + ~2 taps gathered per output")
        if want("A10_alt"):
            alt = S.PytorchAlternateCorrBlock1D(f1[:1], f2[:1], 4, 4)
            add("A10_alt_corr_lookup B=1 [coords scattered up to 30 px in x AND y: every lane gathers from its own rows]",
                timeit(lambda: alt(coords[:1]), 3, 1), None, 2 * 36 * Hr * Wr * Cr * 5 / 1e9, bound="valu")
            # what the loop feeds it (raft_stereo.py:171-180): y is the row itself, x a smooth disparity field
            xs = torch.linspace(0, 6.28, Wr, device=dev).view(1, 1, 1, Wr)
            ys = torch.linspace(0, 3.14, Hr, device=dev).view(1, 1, Hr, 1)
            smooth = S.coords_grid(1, Hr, Wr, device=dev).clone()
            smooth[:, :1] -= 20.0 + 8.0 * torch.sin(xs) * torch.cos(ys)
            smooth = smooth.contiguous()
            add("A10_alt_corr_lookup B=1 [loop-like coords: y = row, smooth disparity]", timeit(lambda: alt(smooth), 3, 1), None,
                2 * 36 * Hr * Wr * Cr * 5 / 1e9, bound="valu")
        if want("A13_upsample"):
            d8 = torch.rand(8, 1, 72, 120, device=dev)
            m8 = torch.randn(8, 576, 72, 120, device=dev)
            nc = 8 * 72 * 120
            add("A13_upsample_convex8 [8,1,72,120]", timeit(lambda: S.upsample_convex8(d8, m8), args.reps),
                (nc * (9 * 64 + 1) * 4 + nc * 64 * 4) / 1e9)
            d4 = torch.rand(8, 2, 136, 240, device=dev)
            m4 = torch.randn(8, 144, 136, 240, device=dev)
            nc = 8 * 136 * 240
            add("A13_upsample_flow f=4 [8,2,136,240]", timeit(lambda: S.upsample_flow(d4, m4, 2), args.reps),
                (nc * (9 * 16 + 2) * 4 + nc * 2 * 16 * 4) / 1e9)
        if want("A12_disp_warp"):
            img = torch.randn(8, 3, 540, 960, device=dev)
            dd = torch.rand(8, 1, 540, 960, device=dev) * 100
            add("A12_disp_warp [8,3,540,960]", timeit(lambda: S.disp_warp(img, dd), args.reps),
                (8 * 540 * 960 * (3 + 1 + 6) * 4) / 1e9)
        if want("gru"):
            h = torch.randn(8, 128, 136, 240, device=dev)
            a, b2 = torch.randn_like(h), torch.randn_like(h)
            add("gru_blend [8,128,136,240] (+2 bias)", timeit(lambda: S.gru_blend(h, a, b2, a, b2), args.reps),
                h.numel() * 4 * 6 / 1e9)
            add("gru_reset [8,128,136,240] (+bias)", timeit(lambda: S.gru_reset(h, a, b2), args.reps), h.numel() * 4 * 4 / 1e9)
        if want("A11_iter"):
            net = S.DisparityUpdateDLCWithMask(24, 32, 20).eval().cuda()
            v = torch.randn(8, 24, 72, 120, device=dev)
            dsp = S.DisparityEstimation(24)(v).unsqueeze(1)
            li = torch.randn(8, 3, 576, 960, device=dev)
            ri = torch.randn(8, 3, 576, 960, device=dev)
            hid = torch.zeros(8, 64, 72, 120, device=dev)
            add("A11 one iteration B=8 [24,72,120] (+A13)",
                timeit(lambda: S.upsample_convex8(*(lambda o: (o[0], o[2]))(net(v, dsp, li, ri, hid, True))), 5), None,
                note="MIOpen convs + HIP pointwise; informational")
        if want("A14_iter"):
            blk14 = S.BasicMultiUpdateBlock_LZH(4, 4, 3, 2, [128, 128, 128]).eval().cuda()
            nets = [torch.tanh(torch.randn(8, 128, Hr >> i, Wr >> i, device=dev)) for i in range(3)]
            inps = [[torch.randn(8, 128, Hr >> i, Wr >> i, device=dev) for _ in range(3)] for i in range(3)]
            corr = torch.randn(8, 36, Hr, Wr, device=dev)
            flow = torch.randn(8, 2, Hr, Wr, device=dev)
            add("A14 one iteration B=8 @136x240 [NCHW: library convolutions + torch.cat]", timeit(lambda: blk14(list(nets), inps, corr, flow), 3, 1), None,
                8 * 168.0, bound="miopen", note="168 GFLOP/iter/pair of convs (MIOpen); informational")
            cl = torch.channels_last
            blk14 = blk14.to(memory_format=cl)
            nets_cl = [t.contiguous(memory_format=cl) for t in nets]
            inps_cl = [[t.contiguous(memory_format=cl) for t in lvl] for lvl in inps]
            add("A14 one iteration B=8 @136x240 [channels-last: split-fp16 conv kernel, gate epilogues, no concatenations]",
                timeit(lambda: blk14(list(nets_cl), inps_cl, corr, flow), 3, 1), None, 8 * 168.0, bound="mfma",
                note="fp32-equivalent flops; frac is against the f32 MFMA peak (157.3): the split-bf16 formulation is not bound by it")
            import torch.nn as nn
            from stereoformer_amd import conv_split as CS
            for (nm, cs, co, relu) in (("convz|convr of gru08: [8,384,136,240] -> 128+128", (128, 128, 128), (128, 128), False),
                                       ("flow_head.conv1|mask[0]: [8,128,136,240] -> 256+256 (+ReLU)", (128,), (256, 256), True),
                                       ("encoder.conv: [8,64+64,136,240] -> 126 (+ReLU)", (64, 64), (126,), True)):
                srcs = [torch.randn(8, c, Hr, Wr, device=dev).contiguous(memory_format=cl) for c in cs]
                convs = [nn.Conv2d(sum(cs), c, 3, padding=1).to(dev).to(memory_format=cl) for c in co]
                gf = 2 * 8 * Hr * Wr * sum(cs) * sum(co) * 9 / 1e9
                ms = timeit(lambda: CS.conv_nhwc(srcs, convs, relu=relu), 10, 3)
                add(f"A14 conv_split {nm}", ms, None, gf, bound="mfma",
                    note=f"{gf / ms / (2500.0 / 6):.3f} of the split formulation's peak (dense bf16 2500 / 6 piece products)")
                xc = torch.cat(srcs, 1)
                add(f"A14 [library, for comparison] {nm}", timeit(lambda: [m(xc) for m in convs], 10, 3), None, gf, bound="mfma",
                    note="MIOpen fp32 implicit GEMM on the pre-concatenated input (torch.cat and ReLU not included)")
                del srcs, convs, xc
            xs = torch.randn(8, 256, Hr, Wr, device=dev).contiguous(memory_format=cl)
            ms_ = nn.Conv2d(256, 2, 3, padding=1).to(dev).to(memory_format=cl)
            add("A14 flow_head.conv2 [8,256,136,240] -> 2 (fp32 vector kernel)", timeit(lambda: CS.conv3x3_small(xs, ms_), 10, 3),
                (8 * 256 * Hr * Wr * 4) / 1e9, 2 * 8 * Hr * Wr * 256 * 2 * 9 / 1e9)
            add("A14 [library, for comparison] flow_head.conv2", timeit(lambda: ms_(xs), 10, 3), (8 * 256 * Hr * Wr * 4) / 1e9)
            del xs, ms_
        if want("bwd"):
            torch.set_grad_enabled(True)
            Lg, Rg = L.detach().requires_grad_(True), R.detach().requires_grad_(True)
            vol_g = S.CostVolume(D)(Lg, Rg)
            cot = torch.randn_like(vol_g)
            from stereoformer_amd import _native as NV
            gl_, gr_ = torch.empty_like(L), torch.empty_like(R)

            def a1_bwd():
                NV.call("stereo_gwc_volume_bwd_f32", NV.ptr(L), NV.ptr(R), NV.ptr(cot), NV.ptr(gl_), NV.ptr(gr_), B, C, H, W, D, 1,
                        NV.stream_ptr())
            add("bwd A1 corr volume cfg-2 (dL + dR)", timeit(a1_bwd, 10, 3),
                (N * D * 4 + 4 * N * C * 4) / 1e9, 2 * 2 * B * H * C * (D * W - D * (D - 1) / 2) / 1e9,
                note="round 3: ONE launch, the gradient volume read once (corr_bwd_fused.hip; split-fp16 MFMA, row-walking "
                     "workgroups); bytes = one read of the volume + both features in + both gradients out")
            with NV.test_options(corr_bwd_fused=0):
                add("bwd A1 [round 2's two launches, for comparison]", timeit(a1_bwd, 5, 2),
                    (2 * N * D * 4 + 4 * N * C * 4) / 1e9, note="each launch reads the gradient volume; f32 MFMA + LDS sheet")
            add("bwd A1 through autograd (CostVolume backward incl. its two output allocations)",
                timeit(lambda: torch.autograd.grad(vol_g, (Lg, Rg), cot, retain_graph=True), 5, 1),
                (N * D * 4 + 4 * N * C * 4) / 1e9)
            del gl_, gr_
            del vol_g, cot
            vd = S.CostVolume(D)(L, R).detach().requires_grad_(True)
            dsp = S.DisparityEstimation(D)(vd)
            cd = torch.randn_like(dsp)
            add("bwd A4 soft-argmin cfg-2",
                timeit(lambda: torch.autograd.grad(dsp, vd, cd, retain_graph=True), 5, 1), (2 * N * D * 4 + N * 4) / 1e9,
                note="1 read + 1 write algorithmic (the second read is cache traffic)")
            d1 = dsp.detach().unsqueeze(1).requires_grad_(True)
            ref = S.LocalCostVolume(2, 20)(vd, d1, True)
            co = torch.randn_like(ref)
            add("bwd A5 local lookup cfg-2 (incl. zero-fill of grad volume)",
                timeit(lambda: torch.autograd.grad(ref, (vd, d1), co, retain_graph=True), 5, 1), (N * D * 4 + 4 * N * 4) / 1e9)
            del vd, dsp, ref
            L1g, R1g = L[:2].detach().requires_grad_(True), R[:2].detach().requires_grad_(True)
            vg = S.build_gwc_volume(L1g, R1g, D, G)
            cg = torch.randn_like(vg)
            n2 = 2 * H * W
            add("bwd A3 gwc volume B=2 G=8 (dL + dR)",
                timeit(lambda: torch.autograd.grad(vg, (L1g, R1g), cg, retain_graph=True), 3, 1),
                (n2 * G * D * 4 + 4 * n2 * C * 4) / 1e9,
                note="round 3: the one-read A1 backward on B G pairs of C / G channels (the same pointers reinterpreted); bytes = ONE "
                     "read of the gradient volume + features in / gradients out (round 2: two reads, 3.5 ms)")
            del vg, cg
            # the backward paths added in round 2 (A6, A2, A7 by composition), one pair
            v1 = S.CostVolume(D)(L[:1], R[:1]).detach().requires_grad_(True)
            d6 = S.DisparityEstimation(D)(v1.detach()).unsqueeze(1).requires_grad_(True)
            o6 = S.Variance_Based_Local_Cost_Volume(1, 20)(v1, d6, True)
            c6 = torch.randn_like(o6)
            n1 = H * W
            add("bwd A6 variance-based lookup, 1 pair (one dense launch)",
                timeit(lambda: torch.autograd.grad(o6, (v1, d6), c6, retain_graph=True), 5, 1), (2 * n1 * D * 4 + 4 * n1 * 4) / 1e9,
                note="round 3 (last day): the gradient is dense in d (variance path), so one launch writes every element once -- "
                     "0.57 ms as thread-per-pixel accumulation into a zero-filled tensor")
            del v1, d6, o6, c6
            La, Ra = L[:1, :8].detach().requires_grad_(True), R[:1, :8].detach().requires_grad_(True)
            for kind in ("difference", "concat"):
                va = S.CostVolume(D, kind)(La, Ra)
                ca = torch.randn_like(va)
                add(f"bwd A2 {kind} volume, 1 pair, 8 channels (dL + dR)",
                    timeit(lambda: torch.autograd.grad(va, (La, Ra), ca, retain_graph=True), 3, 1),
                    (va.numel() * 4 + 4 * n1 * 8 * 4) / 1e9)
                del va, ca
            Lq, Rq = L[:1].detach().requires_grad_(True), R[:1].detach().requires_grad_(True)
            att = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")
            oq, dq = att(Lq, Rq)
            cq, cdq = torch.randn_like(oq), torch.randn_like(dq)
            add("bwd A7 cross-attention under grad (composition of A1 forward / backward kernels), 1 pair fwd+bwd",
                timeit(lambda: torch.autograd.grad(att(Lq, Rq), (Lq, Rq), (cq, cdq)), 3, 1), None,
                note="materialises the [1,D,H,W] score volume (training path; inference uses the fused kernel)")
            del oq, dq, cq, cdq
            # one training step of the cfg-2 chain at B=2 (features -> volume stage -> local lookup -> scalar loss)
            lk = S.LocalCostVolume(2, 20)

            def train_step(fused):
                Lt, Rt = L[:2].detach().requires_grad_(True), R[:2].detach().requires_grad_(True)
                if fused:
                    _, dsp2, mean2 = S.build_gwc_volume_with_disparity(Lt, Rt, D, G, return_mean=True)
                else:
                    vol2 = S.build_gwc_volume(Lt, Rt, D, G)
                    dsp2, mean2 = S.group_mean_soft_argmin(vol2, return_mean=True)
                out2 = lk(mean2, dsp2.unsqueeze(1), True)
                torch.autograd.grad(out2.mean(), (Lt, Rt))

            for fused in (True, False):
                add(f"train step cfg-2 chain B=2 fwd+bwd ({'fused volume stage' if fused else 'composition of the separate ops'})",
                    timeit(lambda: train_step(fused), 3, 1), None,
                    note="features -> group-wise volume + mean + disparity -> LocalCostVolume -> loss; gradients w.r.t. both features")
            torch.set_grad_enabled(False)
        if want("loop_raft"):
            loop = S.RAFTStereoLoop([128, 128, 128]).eval().cuda()
            nets = [torch.tanh(torch.randn(8, 128, Hr >> i, Wr >> i, device=dev)) for i in range(3)]
            inps = [[torch.randn(8, 128, Hr >> i, Wr >> i, device=dev) for _ in range(3)] for i in range(3)]
            for graph in (False, True):
                ms = timeit(lambda: loop(f1, f2, [n.clone() for n in nets], inps, iters=16, test_mode=True,
                                         use_graph=graph), 2, 1)
                add(f"cfg-4 RAFT loop 16 iters B=8 {'graph' if graph else 'eager'}", ms, None, 8 * (15 * 146.3 + 168.0) + 32.09,
                    bound="miopen", note=f"{8 / ms * 1e3:.1f} pairs/s per GPU; conv FLOPs dominate (MIOpen fp32); mask head only in the last iteration")
            loop.channels_last = True     # NHWC weights / activations through the update block (same arithmetic)
            ms = timeit(lambda: loop(f1, f2, [n.clone() for n in nets], inps, iters=16, test_mode=True, use_graph=True), 2, 1)
            add("cfg-4 RAFT loop 16 iters B=8 channels_last graph", ms, None, 8 * (15 * 146.3 + 168.0) + 32.09, bound="miopen",
                note=f"{8 / ms * 1e3:.1f} pairs/s per GPU; fp32, NHWC through the update block")
            loop.mixed_precision = True   # the reference's optional flag: update-block convs under fp16 autocast
            for graph in (False, True):
                ms = timeit(lambda: loop(f1, f2, [n.clone() for n in nets], inps, iters=16, test_mode=True,
                                         use_graph=graph), 2, 1)
                add(f"cfg-4 RAFT loop 16 iters B=8 mixed_precision channels_last {'graph' if graph else 'eager'}", ms, None,
                    8 * (15 * 146.3 + 168.0) + 32.09, bound="miopen",
                    note=f"{8 / ms * 1e3:.1f} pairs/s per GPU; update-block convs fp16 (reference option), rest fp32")
        if want("loop_lowcnn"):
            lp = S.LowCNNGRULoop(24, 32, 20).eval().cuda()
            v = torch.randn(8, 24, 72, 120, device=dev)
            dsp = S.DisparityEstimation(24)(v).unsqueeze(1)
            li = torch.randn(8, 3, 576, 960, device=dev)
            ri = torch.randn(8, 3, 576, 960, device=dev)
            for graph in (False, True):
                ms = timeit(lambda: lp(v, dsp, li, ri, iters=16, use_graph=graph), 3, 1)  # runs under no_grad
                add(f"cfg-4 LowCNN-GRU loop 16 iters B=8 {'graph (last iterate)' if graph else 'eager (all iterates)'}", ms,
                    None, None, note=f"{8 / ms * 1e3:.1f} pairs/s per GPU")
            ms = timeit(lambda: lp(v, dsp, li, ri, iters=16, use_graph=True, graph_all_iterates=True), 3, 1)
            add("cfg-4 LowCNN-GRU loop 16 iters B=8 graph (all iterates)", ms, None, None,
                note=f"{8 / ms * 1e3:.1f} pairs/s per GPU")
            lpc = S.LowCNNGRULoop(24, 32, 20, channels_last=True).eval().cuda()
            ms = timeit(lambda: lpc(v, dsp, li, ri, iters=16, use_graph=True), 3, 1)
            add("cfg-4 LowCNN-GRU loop 16 iters B=8 channels_last graph (last iterate) [3x3 convolutions on conv_split]", ms, None, None,
                note=f"{8 / ms * 1e3:.1f} pairs/s per GPU")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "kernel_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
