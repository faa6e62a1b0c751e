"""Error audit of the update block on the split-operand convolution kernel (every arithmetic form) against the same loop on the library's fp32 convolutions
(dev tool): final low-resolution flow and upsampled disparity after 16 iterations at cfg-4 geometry, two pairs.
    python tools/conv_audit.py > profiles/r02c_conv_error_audit.txt"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import conv_split as C
from tools.bench_configs import raft_inputs
dev = torch.device("cuda")
f1, f2, nets, inps = raft_inputs(2, 136, 240, dev, 400)
torch.manual_seed(4)
loop = S.RAFTStereoLoop([128, 128, 128], channels_last=True).eval().to(dev)
with torch.no_grad():
    for iters in (1, 4, 16):
        lo_a, up_a = loop(f1, f2, [n.clone() for n in nets], inps, iters=iters, test_mode=True)
        real = C.supported, C.small_supported
        C.supported = lambda *a, **k: False
        C.small_supported = lambda *a, **k: False
        lo_b, up_b = loop(f1, f2, [n.clone() for n in nets], inps, iters=iters, test_mode=True)
        C.supported, C.small_supported = real
        for prec, label in (("high", "bf16x3"), ("bf16x6", "split-bf16, six products")):
            loop.conv_precision = prec
            lo_c, up_c = loop(f1, f2, [n.clone() for n in nets], inps, iters=iters, test_mode=True)
            loop.conv_precision = "highest"
            print(f"iterations {iters:2d} [conv_precision='{prec}', {label}]: max |disparity diff| {float((up_c - up_b).abs().max()):.2e}, "
                  f"mean {float((up_c - up_b).abs().mean()):.2e} px", flush=True)
        print(f"iterations {iters:2d} [conv_precision='highest', split-fp16, three products -- the default]: max |flow_lo diff| {float((lo_a - lo_b).abs().max()):.2e} (|flow| max {float(lo_b.abs().max()):.2f}) px at 1/4 scale; "
              f"max |disparity diff| {float((up_a - up_b).abs().max()):.2e}, mean {float((up_a - up_b).abs().mean()):.2e} px at full resolution "
              f"(|disparity| max {float(up_b.abs().max()):.1f})", flush=True)
