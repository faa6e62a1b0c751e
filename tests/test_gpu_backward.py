"""backward of the cfg-2 chain ops (correlation / group-wise volume, soft-argmin (+prob),
fixed-radius local lookup) against gradients produced by the reference's own autograd on CPU (golden G11),
plus finite-difference spot checks.  Tolerances: 2e-5 abs on gradients of O(1) tensors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=2e-5)


def dev(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    return t.requires_grad_(grad)


def close(a, b, **tol):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b, **tol)


@pytest.fixture(scope="module")
def S():
    import stereoformer_amd as s
    return s


def test_volume_backward_g11(S, golden):
    g = golden("g11_backward")
    for (D, G) in ((12, 1), (12, 4), (40, 2)):
        L, R = dev(g["L"], True), dev(g["R"], True)
        vol = S.CostVolume(D)(L, R) if G == 1 else S.build_gwc_volume(L, R, D, G)
        gl, gr = torch.autograd.grad(vol, (L, R), dev(g[f"vol_cot_D{D}_G{G}"]))
        close(gl, g[f"vol_gl_D{D}_G{G}"], **TOL)
        close(gr, g[f"vol_gr_D{D}_G{G}"], **TOL)
    # only one side needs a gradient
    L, R = dev(g["L"], True), dev(g["R"], False)
    (gl,) = torch.autograd.grad(S.build_gwc_volume(L, R, 12, 4), L, dev(g["vol_cot_D12_G4"]))
    close(gl, g["vol_gl_D12_G4"], **TOL)


def test_soft_argmin_backward_g11(S, golden):
    g = golden("g11_backward")
    for sim in (True, False):
        v = dev(g["sa_vol"], True)
        (gv,) = torch.autograd.grad(S.DisparityEstimation(12, sim)(v), v, dev(g["sa_cot_disp"]))
        close(gv, g[f"sa_gvol_sim{int(sim)}"], **TOL)
    v = dev(g["sa_vol"], True)
    d, p = S.DisparityEstimationWithProb(12, True)(v)
    (gv,) = torch.autograd.grad((d, p), v, (dev(g["sa_cot_disp"]), dev(g["sa_cot_prob"])))
    close(gv, g["sa_gvol_withprob"], **TOL)


def test_local_lookup_backward_g11(S, golden):
    g = golden("g11_backward")
    for (r, s) in ((2, 20), (4, 10)):
        for cv in (False, True):
            vol, disp = dev(g["ll_vol"], True), dev(g["ll_disp"], True)
            gv, gd = torch.autograd.grad(S.LocalCostVolume(r, s)(vol, disp, cv), (vol, disp), dev(g["ll_cot"]))
            close(gv, g[f"ll_gvol_r{r}_S{s}_cv{int(cv)}"], rtol=1e-4, atol=5e-5)
            close(gd, g[f"ll_gdisp_r{r}_S{s}_cv{int(cv)}"], rtol=1e-4, atol=5e-5)


def test_upsample_convex_backward_g11(S, golden):
    g = golden("g11_backward")
    d8, m8 = dev(g["up8_disp"], True), dev(g["up8_mask"], True)
    gd, gm = torch.autograd.grad(S.upsample_convex8(d8, m8), (d8, m8), dev(g["up8_cot"]))
    close(gd, g["up8_gdisp"], rtol=1e-4, atol=5e-5)
    close(gm[:, ::31], g["up8_gmask_sub"], rtol=1e-4, atol=5e-5)
    fl, m4 = dev(g["up4_flow"], True), dev(g["up4_mask"], True)
    gf, gm4 = torch.autograd.grad(S.upsample_flow(fl, m4, 2), (fl, m4), dev(g["up4_cot"]))
    close(gf, g["up4_gflow"], rtol=1e-4, atol=5e-5)
    close(gm4, g["up4_gmask"], rtol=1e-4, atol=5e-5)


def test_lowcnn_style_head_trains(S):
    """The reference's fixed-radius LowCNN head (baseline.py:114-135: CostVolume -> aggregation convs ->
    DisparityEstimation -> LocalCostVolume -> ConvAffinityUpsample -> upsample_convex8) trains end to end on
    the HIP path: every custom op on it has a HIP backward."""
    torch.manual_seed(0)
    B, C, h, w, D = 1, 16, 8, 16, 24
    fl = torch.randn(B, C, h, w, device="cuda")
    fr = torch.randn(B, C, h, w, device="cuda")
    agg = torch.nn.Conv2d(D, D, 3, padding=1).cuda()
    up = S.ConvAffinityUpsample(C, 32).cuda()
    params = list(agg.parameters()) + list(up.parameters())
    opt = torch.optim.SGD(params, lr=1e-2)
    gt = torch.full((B, 1, 8 * h, 8 * w), 40.0, device="cuda")
    losses = []
    for _ in range(4):
        opt.zero_grad()
        cost = agg(S.CostVolume(D)(fl, fr) * 4.0)
        disp = S.DisparityEstimation(D)(cost).unsqueeze(1)
        disp = S.LocalCostVolume(2, 10)(cost, disp, True)
        pred = S.upsample_convex8(disp, up(fl))
        loss = (pred - gt).abs().mean()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]


def test_chain_trains_one_step(S):
    """The cfg-2 correlation chain end to end under autograd: loss decreases after one SGD step on the features."""
    torch.manual_seed(0)
    B, C, H, W, D = 1, 8, 6, 64, 24
    L = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    R = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    target = torch.full((B, H, W), 5.0, device="cuda")

    def loss_fn():
        vol = S.CostVolume(D)(L, R) * 8.0
        disp = S.DisparityEstimation(D)(vol)
        refined = S.LocalCostVolume(2, 10)(vol, disp.unsqueeze(1), False).squeeze(1)
        return ((disp - target) ** 2).mean() + ((refined - target) ** 2).mean()

    l0 = loss_fn()
    gl, gr = torch.autograd.grad(l0, (L, R))
    assert torch.isfinite(gl).all() and torch.isfinite(gr).all() and gl.abs().max() > 0
    with torch.no_grad():
        L -= 0.5 * gl
        R -= 0.5 * gr
    assert loss_fn().item() < l0.item()


def test_group_mean_soft_argmin_backward(S):
    """Fused reader's backward == autograd through (mean over groups -> DisparityEstimation) of the HIP ops."""
    torch.manual_seed(2)
    vg = (2.0 * torch.randn(2, 4, 16, 5, 40, device="cuda")).requires_grad_(True)
    cd = torch.randn(2, 5, 40, device="cuda")
    cm = torch.randn(2, 16, 5, 40, device="cuda")
    d, m = S.group_mean_soft_argmin(vg, return_mean=True)
    (g1,) = torch.autograd.grad((d, m), vg, (cd, cm))
    vg2 = vg.detach().clone().requires_grad_(True)
    mean = vg2.mean(1)
    d2 = S.DisparityEstimation(16)(mean)
    (g2,) = torch.autograd.grad((d2, mean), vg2, (cd, cm))
    assert (g1 - g2).abs().max().item() < 2e-6


def test_forward_only_ops_still_refuse_grad(S):
    """Every module surface has a backward since round 2; what stays forward-only are the fused inference entry points and
    the raw candidate sampler -- they refuse a requires_grad input loudly instead of silently detaching it."""
    x = torch.randn(1, 4, 3, 8, device="cuda", requires_grad=True)
    with pytest.raises(RuntimeError, match="forward-only"):
        S.corr_soft_argmin(x, x, 4)
    with pytest.raises(RuntimeError, match="forward-only"):
        S.build_cost_volume_from_volume(x, torch.rand(1, 3, 3, 8, device="cuda"))
    S.CostVolume(2, "difference")(x, x).sum().backward()          # A2 and A6 are differentiable now
    S.Variance_Based_Local_Cost_Volume(1, 5)(x, torch.rand(1, 1, 3, 8, device="cuda")).sum().backward()
    assert torch.isfinite(x.grad).all()


def test_dlc_updater_module_gradients_g11(S, golden):
    """DisparityUpdateDLCWithMask, two chained iterations: gradients w.r.t. the cost volume, the input disparity and
    EVERY parameter against the reference's autograd (convs / BN in MIOpen, all custom ops HIP backward)."""
    from _weights import fill_state_dict_
    g = golden("g11_backward")
    torch.backends.cudnn.allow_tf32 = False
    net = S.DisparityUpdateDLCWithMask(input_channels=24, hidden_dim=8, sample_points=20).eval()
    fill_state_dict_(net, seed=0.5)
    net = net.cuda()
    vol, d0 = dev(g["dlc_vol"], True), dev(g["dlc_disp0"], True)
    left, right = dev(g["dlc_left"]), dev(g["dlc_right"])
    hid = torch.zeros(1, 16, 8, 16, device="cuda")
    dd, hidn, mk = net(vol, d0, left, right, hid, True)
    dd2, hidn2, mk2 = net(vol, dd, left, right, hidn, True)
    cdd, cmk, chid = dev(g["dlc_cdd"]), dev(g["dlc_cmk_full"]), dev(g["dlc_chid"])
    loss = (dd2 * cdd).sum() + (mk2 * cmk).sum() * 0.01 + (hidn2 * chid).sum() + (dd * cdd).sum()
    names = [n for n, _ in net.named_parameters()]
    assert names == list(g["dlc_param_names"])
    grads = torch.autograd.grad(loss, [vol, d0] + [p for _, p in net.named_parameters()])

    def rel(a, b):
        a = a.detach().cpu().numpy()
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))

    assert rel(grads[0], g["dlc_gvol"]) < 2e-3
    assert rel(grads[1], g["dlc_gdisp0"]) < 2e-3
    for n_, g_ in zip(names, grads[2:]):
        ref = g["dlc_gp_" + n_]
        got = g_ if g_.numel() < 20000 else g_.flatten()[::17]
        assert rel(got, ref) < 2e-3, (n_, rel(got, ref))


def test_allpairs_pyramid_and_lookup_backward_vs_torch(S):
    """A8 + A9 backward (pool backward, two f32-MFMA GEMMs, lookup scatter) against torch autograd of the same
    math written with einsum / avg_pool / gather-style sampling on the GPU."""
    import torch.nn.functional as F
    torch.manual_seed(5)
    B, C, H, W1, W2, L, r = 2, 40, 5, 70, 72, 3, 3
    f1 = torch.randn(B, C, H, W1, device="cuda").requires_grad_(True)
    f2 = torch.randn(B, C, H, W2, device="cuda").requires_grad_(True)
    coords = torch.rand(B, 1, H, W1, device="cuda") * (W2 + 8) - 4
    blk = S.CorrBlock1D(f1, f2, num_levels=L, radius=r)
    out = blk(coords)
    cot = torch.randn_like(out)
    g1, g2 = torch.autograd.grad((out * cot).sum(), [f1, f2])

    f1r, f2r = f1.detach().clone().requires_grad_(True), f2.detach().clone().requires_grad_(True)
    corr = torch.einsum("bchi,bchj->bhij", f1r, f2r).reshape(B * H * W1, 1, 1, W2) / C ** 0.5
    pyr = [corr]
    for _ in range(L):
        pyr.append(F.avg_pool2d(pyr[-1], [1, 2], stride=[1, 2]))
    outs = []
    x0 = coords.permute(0, 2, 3, 1).reshape(B * H * W1, 1, 1, 1)
    dx = torch.linspace(-r, r, 2 * r + 1, device="cuda").view(1, 1, 2 * r + 1, 1)
    for i in range(L):
        lvl = pyr[i]
        w = lvl.shape[-1]
        x = (dx + x0 / 2 ** i)[..., 0]                       # [N,1,T]
        xf = torch.floor(x)
        a = x - xf
        v = 0
        for xi, wt in ((xf, 1 - a), (xf + 1, a)):
            ok = ((xi >= 0) & (xi <= w - 1)).float()
            idx = xi.clamp(0, w - 1).long()
            v = v + torch.gather(lvl[:, 0], 2, idx) * wt * ok
        outs.append(v.view(B, H, W1, -1))
    ref = torch.cat(outs, -1).permute(0, 3, 1, 2).contiguous()
    close(out, ref.detach().cpu().numpy(), rtol=1e-4, atol=2e-4)
    r1, r2 = torch.autograd.grad((ref * cot).sum(), [f1r, f2r])
    close(g1, r1.cpu().numpy(), rtol=1e-4, atol=2e-4)
    close(g2, r2.cpu().numpy(), rtol=1e-4, atol=2e-4)


def test_resize_bilinear_backward_vs_torch(S):
    import torch.nn.functional as F
    torch.manual_seed(6)
    for (hi, wi, ho, wo, ac) in ((9, 13, 18, 26, True), (9, 13, 17, 31, False), (16, 20, 8, 10, True), (5, 7, 5, 7, False)):
        x = torch.randn(2, 3, hi, wi, device="cuda").requires_grad_(True)
        y = S.resize_bilinear(x, (ho, wo), ac, scale=2.0)
        cot = torch.randn_like(y)
        (g,) = torch.autograd.grad((y * cot).sum(), x)
        xr = x.detach().clone().requires_grad_(True)
        yr = 2.0 * F.interpolate(xr, (ho, wo), mode="bilinear", align_corners=ac)
        (gr,) = torch.autograd.grad((yr * cot).sum(), xr)
        close(y, yr.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
        close(g, gr.cpu().numpy(), rtol=1e-5, atol=2e-5)


def test_raft_stereo_loop_gradients_g12(S, golden):
    """RAFT_Stereo hot loop, 3 iterations in training mode: gradients w.r.t. fmap1/fmap2, the hidden states, the
    context tensors and EVERY update-block parameter against the reference model's own autograd (golden G12)."""
    from _weights import fill_state_dict_
    g = golden("g12_raft_backward")
    torch.backends.cudnn.allow_tf32 = False
    loop = S.RAFTStereoLoop(hidden_dims=[32, 32, 32]).eval()
    fill_state_dict_(loop.update_block, seed=0.125, keys=list(g["keys"]), prefix="update_block.")
    loop = loop.cuda()
    f1, f2 = dev(g["fmap1"], True), dev(g["fmap2"], True)
    net = [dev(g[f"net{i}"], True) for i in range(3)]
    inp = [[dev(g[f"inp{i}{j}"], True) for j in range(3)] for i in range(3)]
    preds = loop(f1, f2, net, inp, iters=3, test_mode=False)
    loss = sum((p * dev(g[f"cot{k}"])).sum() for k, p in enumerate(preds))
    names = [n for n, _ in loop.update_block.named_parameters()]
    assert names == list(g["param_names"])
    leaves = [f1, f2] + net + [t for lvl in inp for t in lvl]
    grads = torch.autograd.grad(loss, leaves + [p for _, p in loop.update_block.named_parameters()])

    def rel(a, b):
        a = a.detach().cpu().numpy()
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))

    refs = [g["g_fmap1"], g["g_fmap2"]] + [g[f"g_net{i}"] for i in range(3)] + \
           [g[f"g_inp{i}{j}"] for i in range(3) for j in range(3)]
    for k, (got, ref) in enumerate(zip(grads, refs)):
        assert rel(got, ref) < 2e-3, (k, rel(got, ref))
    for n_, g_ in zip(names, grads[len(leaves):]):
        ref = g["gp_" + n_]
        got = g_ if g_.numel() < 20000 else g_.flatten()[::37]
        # MIOpen's weight-gradient kernels sum in a box / algorithm dependent order (2.3e-3 seen on one conv)
        assert rel(got, ref) < 5e-3, (n_, rel(got, ref))


def test_fused_volume_stage_backward_matches_composition(S):
    """build_gwc_volume_with_disparity in grad mode (fused forward, backward through the mean volume) against the
    composition build_gwc_volume -> group_mean_soft_argmin, for losses on the disparity + mean only and on all three."""
    torch.manual_seed(9)
    B, C, H, W, D, G = 2, 16, 5, 72, 40, 4
    L0, R0 = torch.randn(B, C, H, W, device="cuda"), torch.randn(B, C, H, W, device="cuda")
    cd, cm, cv = torch.randn(B, H, W, device="cuda"), torch.randn(B, D, H, W, device="cuda"), torch.randn(B, G, D, H, W, device="cuda")
    for use_vol in (False, True):
        grads = []
        for fused in (True, False):
            L, R = L0.clone().requires_grad_(True), R0.clone().requires_grad_(True)
            if fused:
                vol, disp, mean = S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True)
            else:
                vol = S.build_gwc_volume(L, R, D, G)
                disp, mean = S.group_mean_soft_argmin(vol, return_mean=True)
            loss = (disp * cd).sum() + (mean * cm).sum() + ((vol * cv).sum() if use_vol else 0.0)
            grads.append(torch.autograd.grad(loss, [L, R]))
        for a, b in zip(grads[0], grads[1]):
            close(a, b.cpu().numpy(), rtol=1e-4, atol=5e-5)
    # only the disparity used: gradients still flow, nothing else is materialised
    L = L0.clone().requires_grad_(True)
    _, disp = S.build_gwc_volume_with_disparity(L, R0, D, G)
    (gl,) = torch.autograd.grad((disp * cd).sum(), [L])
    assert torch.isfinite(gl).all() and gl.abs().max().item() > 0


@pytest.mark.parametrize("shape", [(1, 32, 3, 960, 192), (2, 12, 4, 520, 150), (1, 6, 5, 260, 97), (2, 32, 2, 300, 64),
                                   (1, 5, 2, 36, 7), (1, 8, 2, 16, 16), (1, 3, 2, 8, 1), (1, 8, 2, 64, 100)])
def test_corr_volume_backward_mfma_vs_valu_vs_torch(S, shape):
    """A1 backward on the matrix cores (gradient rows read along the sheet's diagonals; dR in mirrored / sheared
    coordinates) against the VALU backward and against torch autograd of the shifted-product formula."""
    B, C, H, W, D = shape
    torch.manual_seed(21)
    L0, R0 = torch.randn(B, C, H, W, device="cuda"), torch.randn(B, C, H, W, device="cuda")
    cot = torch.randn(B, D, H, W, device="cuda")
    got = {}
    for mfma in ("1", "0"):
        with S._native.test_options(corr_mfma=int(mfma)):
            L, R = L0.clone().requires_grad_(True), R0.clone().requires_grad_(True)
            vol = S.CostVolume(D)(L, R)
            got[mfma] = torch.autograd.grad((vol * cot).sum(), [L, R])
    Lr, Rr = L0.clone().requires_grad_(True), R0.clone().requires_grad_(True)
    ref = torch.zeros(B, D, H, W, device="cuda")
    planes = []
    for d in range(D):
        if d < W:
            pl = torch.zeros(B, H, W, device="cuda")
            pl = torch.cat([pl[..., :d], (Lr[..., d:] * Rr[..., :W - d]).mean(1)], -1) if d > 0 else (Lr * Rr).mean(1)
        else:
            pl = torch.zeros(B, H, W, device="cuda")
        planes.append(pl)
    ref = torch.stack(planes, 1)
    gref = torch.autograd.grad((ref * cot).sum(), [Lr, Rr])
    for k in ("1", "0"):
        for a, b in zip(got[k], gref):
            close(a, b.cpu().numpy(), rtol=1e-4, atol=5e-5)


def _corr_bwd_fp64(L, R, cot):
    """dL, dR of vol[b, d, h, w] = mean_c L[b, c, h, w] R[b, c, h, w - d] in fp64 (plain shifted products)."""
    B, C, H, W = L.shape
    D = cot.shape[1]
    L, R, cot = L.double(), R.double(), cot.double()
    dL, dR = torch.zeros_like(L), torch.zeros_like(R)
    for d in range(min(D, W)):
        g = cot[:, d:d + 1, :, d:] / C
        dL[..., d:] += g * R[..., :W - d]
        dR[..., :W - d] += g * L[..., d:]
    return dL, dR


@pytest.mark.parametrize("shape", [(1, 32, 3, 960, 192), (2, 12, 4, 520, 150), (1, 5, 3, 388, 180), (2, 8, 200, 260, 40),
                                   (1, 32, 2, 128, 33), (1, 3, 2, 8, 1), (1, 7, 600, 132, 70), (1, 32, 2, 1216, 192)])
@pytest.mark.parametrize("gscale", [1.0, 3e-9, 2e7])
def test_corr_volume_backward_single_read_kernel(S, shape, gscale):
    """Round 3: both A1 feature gradients from ONE pass over the gradient volume (corr_bwd_fused.hip: row-walking
    workgroups, dR tiles carried across rectangles, split-fp16 products with a per-output-column power-of-two scale on the
    gradient tile) against fp64 and against the two-launch f32-MFMA kernel, for gradient magnitudes far outside fp16's
    range (a mean-reduced loss hands over 1e-9-sized gradients).  Shapes: several rectangles per row with a ragged last
    one, D not a multiple of 32 and at the 192 limit, C < 32, more rows than CUs (workgroups chain rows), tiny W."""
    B, C, H, W, D = shape
    torch.manual_seed(33)
    L0, R0 = torch.randn(B, C, H, W, device="cuda"), torch.randn(B, C, H, W, device="cuda")
    cot = torch.randn(B, D, H, W, device="cuda") * gscale
    got = {}
    for fused in (1, 0):
        with S._native.test_options(corr_bwd_fused=fused):
            L, R = L0.clone().requires_grad_(True), R0.clone().requires_grad_(True)
            vol = S.CostVolume(D)(L, R)
            got[fused] = [g.double() / gscale for g in torch.autograd.grad(vol, [L, R], cot)]
    ref = [g / gscale for g in _corr_bwd_fp64(L0, R0, cot)]
    bound = 0.0
    for k in (1, 0):
        for a, b in zip(got[k], ref):
            err = (a - b).abs().max().item()
            assert err <= 2e-5, (k, err)      # fp32 accumulation over <= 256 x 32 products of N(0,1) values / C
            bound = max(bound, err)
    # the split-fp16 kernel is as close to fp64 as the f32 matrix cores, up to the 2^-22 of a product of two 22-bit operands
    e1 = max((a - b).abs().max().item() for a, b in zip(got[1], ref))
    e0 = max((a - b).abs().max().item() for a, b in zip(got[0], ref))
    assert e1 <= 2.0 * e0 + 5e-7, (e1, e0)


def test_corr_volume_backward_single_read_kernel_random_shapes(S):
    """Twelve random shapes inside the one-read kernel's envelope (C <= 32, D <= 192, W % 4 == 0; rows per workgroup, ragged
    rectangles, partial last blocks all vary) against fp64."""
    rng = np.random.default_rng(2024)
    for _ in range(12):
        B = int(rng.integers(1, 3))
        C = int(rng.integers(1, 33))
        H = int(rng.integers(1, 400)) if rng.random() < 0.3 else int(rng.integers(1, 9))
        W = 4 * int(rng.integers(1, 330 if H < 20 else 40))
        D = int(rng.integers(1, 193))
        # feature magnitudes from 0.004 to 250 (the low fp16 piece of a small feature is subnormal: absolute accuracy)
        L0 = torch.randn(B, C, H, W, device="cuda") * float(rng.choice([0.004, 1.0, 250.0]))
        R0 = torch.randn(B, C, H, W, device="cuda") * float(rng.choice([0.004, 1.0, 250.0]))
        cot = torch.randn(B, D, H, W, device="cuda") * float(10.0 ** rng.integers(-6, 4))
        L, R = L0.clone().requires_grad_(True), R0.clone().requires_grad_(True)
        got = torch.autograd.grad(S.CostVolume(D)(L, R), [L, R], cot)
        ref = _corr_bwd_fp64(L0, R0, cot)
        for a, b in zip(got, ref):
            scale = max(b.abs().max().item(), 1e-30)
            assert (a.double() - b).abs().max().item() <= 2e-5 * scale, (B, C, H, W, D)


@pytest.mark.parametrize("D", [7, 48, 96, 150, 192])
def test_soft_argmin_backward_with_addend(S, D):
    """stereo_soft_argmin_bwd_add_f32: grad_vol = addend + soft-argmin backward, bit for bit what the plain entry followed by
    an fp32 add gives up to the one fused multiply-add (<= 1 ulp of the sum); in place (grad_vol == addend) as well."""
    from stereoformer_amd import _native as N
    torch.manual_seed(D)
    B, H, W = 2, 6, 52
    vol = torch.randn(B, D, H, W, device="cuda") * 3
    gd = torch.randn(B, H, W, device="cuda")
    add = torch.randn(B, D, H, W, device="cuda")
    plain, fused = torch.empty_like(vol), torch.empty_like(vol)
    N.call("stereo_soft_argmin_bwd_f32", N.ptr(vol), N.ptr(gd), N.ptr(None), N.ptr(plain), B, D, H, W, 0, N.stream_ptr())
    N.call("stereo_soft_argmin_bwd_add_f32", N.ptr(vol), N.ptr(gd), N.ptr(add), N.ptr(fused), B, D, H, W, 0, N.stream_ptr())
    ref = plain.double() + add.double()
    assert (fused.double() - ref).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item())
    inplace = add.clone()
    N.call("stereo_soft_argmin_bwd_add_f32", N.ptr(vol), N.ptr(gd), N.ptr(inplace), N.ptr(inplace), B, D, H, W, 0, N.stream_ptr())
    assert torch.equal(inplace, fused)
    # outside the envelope the entry says so (the autograd node then adds the cotangents itself)
    with pytest.raises(N.StereoHipError) as ei:
        N.call("stereo_soft_argmin_bwd_add_f32", N.ptr(vol), N.ptr(gd), N.ptr(add), N.ptr(fused), B, D, 3, 3, 0, N.stream_ptr())
    assert ei.value.code == N.ERR_UNSUPPORTED


# ------------------------------------------------------------------ round 2: A6, A2, A7
def test_variance_lookup_backward_g15(S, golden):
    """A6 (Variance_Based_Local_Cost_Volume) gradients w.r.t. the volume and the disparity against the reference's own
    autograd (golden G15): three (gamma, S) settings x both validity modes, incl. edge pixels and a sharply peaked column."""
    g = golden("g15_backward2")
    for (gm, Sp) in ((1.0, 10), (2.0, 20), (0.5, 7)):
        for cv in (False, True):
            vol = dev(g["vl_vol"]).requires_grad_(True)
            disp = dev(g["vl_disp"]).requires_grad_(True)
            out = S.Variance_Based_Local_Cost_Volume(gm, Sp)(vol, disp, cv)
            close(out, g[f"vl_out_g{gm}_S{Sp}_cv{int(cv)}"], rtol=1e-4, atol=2e-4)
            gv, gd = torch.autograd.grad(out, (vol, disp), dev(g["vl_cot"]))
            close(gv, g[f"vl_gvol_g{gm}_S{Sp}_cv{int(cv)}"], rtol=2e-3, atol=2e-4)
            close(gd, g[f"vl_gdisp_g{gm}_S{Sp}_cv{int(cv)}"], rtol=2e-3, atol=5e-4)


@pytest.mark.parametrize("shape", [(1, 24, 5, 64), (2, 48, 3, 100), (1, 192, 4, 132), (2, 100, 7, 36), (1, 33, 1, 4), (1, 7, 2, 260)])
@pytest.mark.parametrize("cv", [False, True])
def test_variance_lookup_backward_dense_launch(S, shape, cv):
    """A6 backward in one dense launch (no zero-fill; every element of grad_vol written once) against the accumulate-into-zeros
    kernel it replaces: same terms, the column statistics reduced in another order -- peaked and flat columns, disparities at
    both ends of the range, every register-slice size (D 7 .. 192), workgroups with partial pixel blocks."""
    from stereoformer_amd import _native as N
    B, D, H, W = shape
    torch.manual_seed(D * 7 + W)
    vol = torch.randn(B, D, H, W, device="cuda") * 3
    vol[:, D // 3, :, : W // 2] += 12.0                          # a sharply peaked half
    disp = torch.rand(B, 1, H, W, device="cuda") * (D - 1)
    disp[..., :2] = 0.0
    disp[..., -2:] = D - 1
    cot = torch.randn(B, 1, H, W, device="cuda")
    for (gm, Sp) in ((1.0, 10), (2.0, 20), (0.5, 7)):
        dense, gd1 = torch.full_like(vol, float("nan")), torch.empty_like(disp)
        acc, gd0 = torch.zeros_like(vol), torch.empty_like(disp)
        N.call("stereo_variance_lookup_bwd_dense_f32", N.ptr(vol), N.ptr(disp), N.ptr(cot), N.ptr(dense), N.ptr(gd1), B, D, H, W, gm,
               Sp, int(cv), N.stream_ptr(vol.device))
        N.call("stereo_variance_lookup_bwd_f32", N.ptr(vol), N.ptr(disp), N.ptr(cot), N.ptr(acc), N.ptr(gd0), B, D, H, W, gm, Sp,
               int(cv), N.stream_ptr(vol.device))
        assert torch.isfinite(dense).all()
        scale = max(acc.abs().max().item(), 1e-6)
        # (a sharply peaked column has a tiny sigma, and 1 / (2 sigma) amplifies the rounding of the variance sum: 1e-4-level
        # differences between two summation orders are fp32 conditioning, not a discrepancy; and where every candidate of a
        # pixel is the current disparity (invalid pixels) the true tap gradient is 0 and both kernels return rounding noise of
        # |g| D eps)
        assert (dense - acc).abs().max().item() <= 2e-4 * scale + 1e-4, (gm, Sp)
        assert (gd1 - gd0).abs().max().item() <= 5e-4 * max(gd0.abs().max().item(), 1e-6) + 1e-4, (gm, Sp)
    with pytest.raises(N.StereoHipError):                        # outside the envelope: refused, the host takes the other entry
        N.call("stereo_variance_lookup_bwd_dense_f32", N.ptr(vol), N.ptr(disp), N.ptr(cot), N.ptr(dense), N.ptr(gd1), B, D, H, W, 1.0,
               21, int(cv), N.stream_ptr(vol.device))


def test_shift_volume_backward_g15(S, golden):
    """A2 'difference' / 'concat' volume gradients against the reference's autograd (G15), D in {1, 7, W, > W}; exact sums of
    the same terms in another order."""
    g = golden("g15_backward2")
    for kind in ("difference", "concat"):
        for D in (1, 7, 20, 26):
            L = dev(g["sv_L"]).requires_grad_(True)
            R = dev(g["sv_R"]).requires_grad_(True)
            v = S.CostVolume(D, kind)(L, R)
            gl, gr = torch.autograd.grad(v, (L, R), dev(g[f"sv_cot_{kind}_D{D}"]))
            close(gl, g[f"sv_gl_{kind}_D{D}"], rtol=1e-5, atol=1e-5)
            close(gr, g[f"sv_gr_{kind}_D{D}"], rtol=1e-5, atol=1e-5)
    L = dev(g["sv_L"]).requires_grad_(True)       # one-sided: only the left features need a gradient
    v = S.build_concat_volume(L, dev(g["sv_R"]), 7)
    (gl,) = torch.autograd.grad(v, (L,), dev(g["sv_cot_concat_D7"]))
    close(gl, g["sv_gl_concat_D7"], rtol=1e-5, atol=1e-5)


def _attention_reference(L, R, V, D, mask, scale):
    """Plain-torch fp64 statement of the build-defined cross-attention (A7 has no reference body: identity-pinned)."""
    B, C, H, W = L.shape
    sc = 1.0 / C if scale == "mean" else 1.0 / C ** 0.5
    logits, vals = [], []
    for d in range(D):
        Rs = torch.zeros_like(R)
        Vs = torch.zeros_like(V)
        if d < W:
            Rs[..., d:] = R[..., :W - d]
            Vs[..., d:] = V[..., :W - d]
        lg = (L * Rs).sum(1) * sc
        if mask == "neginf":
            w = torch.arange(W, device=L.device).view(1, 1, W)
            lg = lg.masked_fill(w < d, float("-inf"))
        logits.append(lg)
        vals.append(Vs)
    P = torch.softmax(torch.stack(logits, 1), 1)                  # [B,D,H,W]
    disp = (P * torch.arange(D, device=L.device, dtype=L.dtype).view(1, D, 1, 1)).sum(1)
    out = sum(P[:, d:d + 1] * vals[d] for d in range(D))
    return out, disp


@pytest.mark.parametrize("cfg", [(2, 8, 3, 40, 12, 8), (1, 32, 2, 64, 20, 5), (1, 48, 2, 36, 9, 48)])
@pytest.mark.parametrize("mask", ["zero", "neginf"])
@pytest.mark.parametrize("scale", ["sqrt_c", "mean"])
def test_cross_attention_backward(S, cfg, mask, scale):
    """A7 under grad: the differentiable composition (A1 forward / backward kernels for Q K^T and P V) reproduces the fused
    inference kernel's outputs and the gradients of a plain fp64 torch statement of the same function w.r.t. left, right
    and value features."""
    B, C, H, W, D, Cv = cfg
    torch.manual_seed(33)
    L0, R0 = torch.randn(B, C, H, W, device="cuda"), torch.randn(B, C, H, W, device="cuda")
    V0 = torch.randn(B, Cv, H, W, device="cuda")
    att = S.CrossAttentionCostVolume(D, mask, scale)
    with torch.no_grad():
        o_inf, d_inf = att(L0, R0, V0)
    L, R, V = (t.clone().requires_grad_(True) for t in (L0, R0, V0))
    o, d = att(L, R, V)
    assert (o - o_inf).abs().max().item() < 5e-5 and (d - d_inf).abs().max().item() < 1e-3
    co, cd = torch.randn_like(o), torch.randn_like(d)
    gl, gr, gv = torch.autograd.grad((o * co).sum() + (d * cd).sum(), (L, R, V))
    Lr, Rr, Vr = (t.double().clone().requires_grad_(True) for t in (L0, R0, V0))
    orf, drf = _attention_reference(Lr, Rr, Vr, D, mask, scale)
    rl, rr, rv = torch.autograd.grad((orf * co.double()).sum() + (drf * cd.double()).sum(), (Lr, Rr, Vr))
    for a, b in ((gl, rl), (gr, rr), (gv, rv)):
        close(a, b.float().cpu().numpy(), rtol=2e-3, atol=3e-4)
    # value defaults to the right features: its gradient is the sum of both roles
    R2 = R0.clone().requires_grad_(True)
    o2, d2 = att(L0, R2)
    (g2,) = torch.autograd.grad(o2.sum() + (d2 * cd).sum(), (R2,))
    Rr2 = R0.double().clone().requires_grad_(True)
    or2, dr2 = _attention_reference(L0.double(), Rr2, Rr2, D, mask, scale)
    (r2,) = torch.autograd.grad(or2.sum() + (dr2 * cd.double()).sum(), (Rr2,))
    close(g2, r2.float().cpu().numpy(), rtol=2e-3, atol=3e-4)


def test_alternate_corr_block_under_grad(S):
    """A10 under grad computes through the all-pairs pyramid (identity I7): same gradients as CorrBlock1D, forward within
    1e-4 of the on-the-fly kernel; a y coordinate off the pixel's own row is refused."""
    torch.manual_seed(9)
    B, C, H, W = 1, 16, 3, 40
    f1, f2 = torch.randn(B, C, H, W, device="cuda"), torch.randn(B, C, H, W, device="cuda")
    coords = S.coords_grid(B, H, W, device="cuda")
    coords[:, 0] -= torch.rand(B, H, W, device="cuda") * 6
    with torch.no_grad():
        ref = S.PytorchAlternateCorrBlock1D(f1, f2, 4, 4)(coords)
    a1, a2 = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    out = S.PytorchAlternateCorrBlock1D(a1, a2, 4, 4)(coords)
    assert (out - ref).abs().max().item() < 1e-4
    cot = torch.randn_like(out)
    ga = torch.autograd.grad(out, (a1, a2), cot)
    b1, b2 = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    gb = torch.autograd.grad(S.CorrBlock1D(b1, b2, 4, 4)(coords), (b1, b2), cot)
    for x, y in zip(ga, gb):
        assert torch.equal(x, y)
    bad = coords.clone()
    bad[:, 1] += 0.25
    with pytest.raises(RuntimeError, match="own row"):
        S.PytorchAlternateCorrBlock1D(a1, a2, 4, 4)(bad)


@pytest.mark.parametrize("D", [7, 48, 100, 192])
def test_soft_argmin_backward_single_pass_vs_two_pass_vs_torch(S, D):
    """A4's backward with the softmax column held in registers (one read of the volume, one write of the gradient) against the
    two-pass kernel (test hook) and torch's own autograd of softmax-expectation, both similarity signs, ragged last block."""
    torch.manual_seed(700 + D)
    B, H, W = 2, 5, 52
    for sim in (True, False):
        v = (torch.randn(B, D, H, W, device="cuda") * 3).requires_grad_(True)
        cot = torch.randn(B, H, W, device="cuda")
        (g1,) = torch.autograd.grad(S.DisparityEstimation(D, sim)(v), v, cot)
        with S._native.test_options(prob_single_pass=0):
            (g2,) = torch.autograd.grad(S.DisparityEstimation(D, sim)(v), v, cot)
        x = v.detach().double().requires_grad_(True)
        p = torch.softmax(x if sim else -x, 1)
        ref = (p * torch.arange(D, device="cuda", dtype=torch.float64).view(1, D, 1, 1)).sum(1)
        (g3,) = torch.autograd.grad(ref, x, cot.double())
        assert float((g1 - g2).abs().max()) <= 5e-6 * max(1.0, float(g2.abs().max()))   # (fp32 summation order of the expectation)
        assert float((g1.double() - g3).abs().max()) <= 2e-5 * max(1.0, float(g3.abs().max()))


@pytest.mark.parametrize("cfg", [(2, 24, 6, 20, 2.0, 20, True), (1, 48, 3, 52, 4.0, 10, False), (2, 12, 5, 8, 6.0, 7, True),
                                 (1, 30, 7, 36, 0.0, 3, False), (1, 20, 3, 7, 2.0, 20, True), (1, 40, 2, 12, 9.0, 12, True)])
def test_local_lookup_backward_dense_vs_accumulate(S, cfg):
    """LocalCostVolume's backward writes the gradient volume densely in one launch (2 r + 3 <= 16, H*W % 4 == 0); the zero-fill +
    accumulate kernel serves the rest: both must agree (same per-pixel arithmetic), incl. disparities at / beyond the volume's
    ends and both validity modes."""
    from stereoformer_amd import _native as N
    B, D, H, W, radius, Sp, cv = cfg
    torch.manual_seed(800 + D)
    vol = torch.randn(B, D, H, W, device="cuda") * 2
    disp = torch.rand(B, 1, H, W, device="cuda") * (D + 4) - 2
    go = torch.randn(B, 1, H, W, device="cuda")
    a, ad = torch.zeros_like(vol), torch.empty_like(disp)
    N.call("stereo_local_lookup_bwd_f32", N.ptr(vol), N.ptr(disp), N.ptr(go), N.ptr(a), N.ptr(ad), B, D, H, W, radius, Sp, int(cv),
           N.stream_ptr())
    v = vol.clone().requires_grad_(True)
    d = disp.clone().requires_grad_(True)
    gv, gd = torch.autograd.grad(S.LocalCostVolume(radius, Sp)(v, d, cv), (v, d), go)
    assert float((gv - a).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max())) and float((gd - ad).abs().max()) <= 1e-6 * max(1.0, float(ad.abs().max()))
    if 2 * radius + 3 <= 16 and (H * W) % 4 == 0:      # the dense kernel took it: check it directly too
        b_, bd = torch.full_like(vol, float("nan")), torch.empty_like(disp)
        N.call("stereo_local_lookup_bwd_dense_f32", N.ptr(vol), N.ptr(disp), N.ptr(go), N.ptr(b_), N.ptr(bd), B, D, H, W, radius, Sp,
               int(cv), N.stream_ptr())
        assert torch.isfinite(b_).all() and float((b_ - a).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max()))
    else:
        with pytest.raises(N.StereoHipError):
            N.call("stereo_local_lookup_bwd_dense_f32", N.ptr(vol), N.ptr(disp), N.ptr(go), N.ptr(a), N.ptr(ad), B, D, H, W, radius,
                   Sp, int(cv), N.stream_ptr())
