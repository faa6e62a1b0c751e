"""dev: error pattern of one cross-attention case per kernel hook"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
import oracle
B, C, H, W, D, Cv = 1, 48, 2, 72, 33, 20
rng = np.random.default_rng(73)
def h16(x):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda().half()
    return t, t.float().cpu().numpy()
Lh, L = h16(rng.standard_normal((B, C, H, W), dtype=np.float32))
Rh, R = h16(rng.standard_normal((B, C, H, W), dtype=np.float32))
Vh, V = h16(rng.standard_normal((B, Cv, H, W), dtype=np.float32))
oo, od = oracle.cross_attention(L, R, V, D, "zero", "sqrt_c")
for half in (True, False):
    for kern in (0, 6, 8):
        with S._native.test_options(att_kernel=kern):
            a = (Lh, Rh, Vh) if half else (Lh.float(), Rh.float(), Vh.float())
            o, d = S.CrossAttentionCostVolume(D, "zero", "sqrt_c")(*a)
        e = np.abs(o.cpu().numpy() - oo)
        print("half", half, "kern", kern, "max err", e.max(), "disp err", np.abs(d.cpu().numpy() - od).max())
        print(" per channel", np.round(e.max(axis=(0, 2, 3)), 3))
        print(" per row", np.round(e.max(axis=(0, 1, 3)), 3))
        print(" per w", np.round(e.max(axis=(0, 1, 2)), 2))
