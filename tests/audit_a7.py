"""Error audit of the split-bf16 cross-attention kernels against the CPU oracle (a script, not a collected test; it lives under
tests/ because only tests may use the oracle):
per config, max / mean |disparity error| and max |output error| for fp32 and fp16 features, both mask modes.
    python tests/audit_a7.py > profiles/r02c_a7_error_audit.txt"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
import oracle
torch.manual_seed(0)
print("config | features | mask | kernel | max |d err| px | mean |d err| px | max |out err| (|out| max)")
for (nm, B, C, H, W, D, rows, scale) in (("cfg-3  [*,32,540,960] D=192", 1, 32, 540, 960, 192, 24, 1.0),
                                         ("cfg-3b [*,256,136,240] D=48", 1, 256, 136, 240, 48, 16, 0.25),
                                         ("cfg-5  [*,256,96,312] D=64", 1, 256, 96, 312, 64, 16, 0.25)):
    L = torch.randn(B, C, H, W, device="cuda") * scale
    R = torch.randn(B, C, H, W, device="cuda") * scale
    for half in (False, True):
        l, r = (L.half(), R.half()) if half else (L, R)
        ln, rn = l.float().cpu().numpy()[:, :, :rows].copy(), r.float().cpu().numpy()[:, :, :rows].copy()
        for mask in ("zero", "neginf"):
            with torch.no_grad():
                o, d = S.CrossAttentionCostVolume(D, mask, "sqrt_c")(l, r)
            oo, od = oracle.cross_attention(ln, rn, None, D, mask, "sqrt_c")
            de = np.abs(d.cpu().numpy()[:, :rows] - od)
            oe = np.abs(o.cpu().numpy()[:, :, :rows] - oo)
            kern = "split-bf16 online (v7)" if C <= 32 else "split-bf16 chunked (v8)"
            print(f"{nm} | {'fp16' if half else 'fp32'} | {mask} | {kern} | {de.max():.2e} | {de.mean():.2e} | {oe.max():.2e} ({np.abs(oo).max():.2f})",
                  flush=True)
