"""A few cross-attention launches for PMC passes (dev tool): python tools/a7_one.py [cfg5] [fp16] [disp]
default: cfg-3 [8,32,540,960] D=192, out + disparity; cfg5: [4,256,96,312] D=64, disparity only (what the cfg-5 chain runs)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
if "cfg5" in sys.argv:
    B, C, H, W, D, mask, need_out = 4, 256, 96, 312, 64, "neginf", False
else:
    B, C, H, W, D, mask, need_out = 8, 32, 540, 960, 192, "zero", "disp" not in sys.argv
L = torch.randn(B, C, H, W, device="cuda"); R = torch.randn(B, C, H, W, device="cuda")
if "fp16" in sys.argv: L, R = L.half(), R.half()
att = S.CrossAttentionCostVolume(D, mask, "sqrt_c")
with torch.no_grad():
    for _ in range(4): att(L, R, need_out=need_out)
torch.cuda.synchronize()
