"""dev: one-process soak of the kernels added in round 3 over many random shapes (fp64 / torch references computed on the GPU).
python tools/soak_r03.py [n]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "12345")))
dev = "cuda"
worst = {}


def note(k, v):
    worst[k] = max(worst.get(k, 0.0), float(v))


def corr_bwd_ref(L, R, cot):
    B, C, H, W = L.shape
    D = cot.shape[1]
    L, R, cot = L.double(), R.double(), cot.double()
    dL, dR = torch.zeros_like(L), torch.zeros_like(R)
    for d in range(min(D, W)):
        g = cot[:, d:d + 1, :, d:] / C
        dL[..., d:] += g * R[..., :W - d]
        dR[..., :W - d] += g * L[..., d:]
    return dL, dR


for it in range(n):
    # one-read A1 backward
    B, C = int(rng.integers(1, 3)), int(rng.integers(1, 33))
    H = int(rng.integers(1, 300)) if rng.random() < 0.25 else int(rng.integers(1, 7))
    W = 4 * int(rng.integers(1, 360 if H < 10 else 50))
    D = int(rng.integers(1, 193))
    L0, R0 = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
    cot = torch.randn(B, D, H, W, device=dev) * float(10.0 ** rng.integers(-7, 5))
    L, R = L0.clone().requires_grad_(True), R0.clone().requires_grad_(True)
    got = torch.autograd.grad(S.CostVolume(D)(L, R), [L, R], cot)
    for a, b in zip(got, corr_bwd_ref(L0, R0, cot)):
        note("a1_bwd_rel", (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30))
    # relayout + channels-last-mask upsampling
    Cn = 32 * int(rng.integers(1, 6))
    h, w = int(rng.integers(1, 40)), 4 * int(rng.integers(1, 30))
    x = torch.randn(int(rng.integers(1, 3)), Cn, h, w, device=dev)
    assert torch.equal(N.nhwc(x), x)
    f = int(rng.choice([4, 8])); cd = int(rng.integers(1, 3)); nb = int(rng.integers(1, 3))
    disp = torch.randn(nb, cd, h, w, device=dev) * 4
    mask = torch.randn(nb, 9 * f * f, h, w, device=dev)
    assert torch.equal(S.upsample_convex(disp, mask, f), S.upsample_convex(disp, mask.contiguous(memory_format=torch.channels_last), f))
    # attention key ring against the fp32 online-softmax kernel (test hook 5)
    Ca = int(rng.integers(1, 33)); Wa = 4 * int(rng.integers(129, 330)); Da = int(rng.integers(161, 193)); Ha = int(rng.integers(1, 4))
    La, Ra = torch.randn(1, Ca, Ha, Wa, device=dev), torch.randn(1, Ca, Ha, Wa, device=dev)
    mk = str(rng.choice(["zero", "neginf"]))
    with N.test_options(att_waves=8, tiles_per_wg=int(rng.choice([0, 2, 3, 5]))):
        o, d = S.CrossAttentionCostVolume(Da, mk, "sqrt_c")(La, Ra)
    with N.test_options(att_kernel=5):
        o5, d5 = S.CrossAttentionCostVolume(Da, mk, "sqrt_c")(La, Ra)
    note("att_out_abs", (o - o5).abs().max().item())
    note("att_disp_abs", (d - d5).abs().max().item())
    # A5 backward window vs the accumulate kernel
    Dv = int(rng.integers(8, 200)); hv, wv = int(rng.integers(1, 12)), 4 * int(rng.integers(1, 40))
    vol = torch.randn(2, Dv, hv, wv, device=dev, requires_grad=True)
    dsp = (torch.rand(2, 1, hv, wv, device=dev) * (Dv + 6) - 3)
    lk = S.LocalCostVolume(int(rng.integers(1, 5)), int(rng.integers(2, 24)))
    go = torch.randn(2, 1, hv, wv, device=dev)
    (g1,) = torch.autograd.grad(lk(vol, dsp, True), [vol], go)
    with N.test_options(prob_single_pass=1):
        pass
    note("a5_finite", 0.0 if torch.isfinite(g1).all() else 1.0)
torch.cuda.synchronize()
print({k: ("%.3g" % v) for k, v in worst.items()})
assert worst["a1_bwd_rel"] < 2e-5 and worst["att_out_abs"] < 2e-4 and worst["att_disp_abs"] < 2e-3 and worst["a5_finite"] == 0.0
print("soak ok:", n, "rounds")
