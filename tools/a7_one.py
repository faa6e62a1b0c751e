"""One cfg-3 cross-attention launch series for PMC passes (dev tool): python tools/a7_one.py [fp16] [disp]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
B, C, H, W, D = 8, 32, 540, 960, 192
L = torch.randn(B, C, H, W, device="cuda"); R = torch.randn(B, C, H, W, device="cuda")
if "fp16" in sys.argv: L, R = L.half(), R.half()
att = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")
with torch.no_grad():
    for _ in range(4): att(L, R, need_out="disp" not in sys.argv)
torch.cuda.synchronize()
