#!/usr/bin/env python3
"""(A script, not a collected test; under tests/ because only tests may use the oracle.)
Error of the soft-argmin family against an fp64 ground truth (numpy), next to the fp32 C oracle's
own error: tells summation-order noise from exp() approximation error."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle, stereoformer_amd as S

def truth(vol):
    v = vol.astype(np.float64)
    v = v - v.max(1, keepdims=True)
    p = np.exp(v); p /= p.sum(1, keepdims=True)
    return (p * np.arange(vol.shape[1])[None, :, None, None]).sum(1)

rng = np.random.default_rng(0)
for D, amp in ((64, 1.0), (192, 1.0), (192, 4.0), (352, 1.0)):
    vol = (amp * rng.standard_normal((1, D, 16, 256))).astype(np.float32)
    t = truth(vol)
    g = S.DisparityEstimation(D)(torch.from_numpy(vol).cuda()).cpu().numpy()
    o = oracle.soft_argmin(vol)
    tt = torch.softmax(torch.from_numpy(vol), 1)
    tr = (tt * torch.arange(D).view(1, D, 1, 1)).sum(1).numpy()
    print(f"D={D} amp={amp}: gpu mean {np.abs(g-t).mean():.2e} max {np.abs(g-t).max():.2e} | oracle mean {np.abs(o-t).mean():.2e} "
          f"max {np.abs(o-t).max():.2e} | torch-cpu mean {np.abs(tr-t).mean():.2e} max {np.abs(tr-t).max():.2e}")
