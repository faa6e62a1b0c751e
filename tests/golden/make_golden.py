#!/usr/bin/env python3
"""Generate golden fixtures by running the REAL reference on CPU.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  Writes small ``tests/golden/*.npz`` files holding seeded
inputs and the reference's outputs; the reference's source is only imported,
never copied.  Usage:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 /root/repo/tests/golden/make_golden.py

Fixture groups follow SURVEY.md section 8(c): G1 volumes (A1/A2/A3), G2 soft-argmin
(A4), G3 local lookups (A5/A6), G4 RAFT correlation (A8/A9/A10), G5 GRU local
updater (A11), G6 disp_warp (A12), G7 convex upsample (A13), G8 RAFT update
block (A14), G9 the RAFT_Stereo hot loop end to end (A8+A9+A13+A14+A15), G13 the same loop started from `flow_init`
plus InputPadder values.
"""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = os.environ.get("STEREO_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.nn.functional as F

from _weights import fill_state_dict_

torch.set_grad_enabled(False)


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# ------------------------------------------------------------------ G1
def g1_volumes():
    from models.utils.build_cost_volume import CostVolume
    from models.CrossAttentionCostVolume.GWC_Cost_Volume import build_gwc_volume

    torch.manual_seed(101)
    L = torch.randn(2, 16, 5, 40)
    R = torch.randn(2, 16, 5, 40)
    out = {"L": npy(L), "R": npy(R)}
    for D in (1, 12, 40):
        out[f"corr_D{D}"] = npy(CostVolume(D, "correlation")(L, R))
    for (D, G) in ((1, 4), (12, 1), (12, 4), (12, 16), (40, 2)):
        out[f"gwc_D{D}_G{G}"] = npy(build_gwc_volume(L, R, D, G))
    for D in (1, 12):
        out[f"diff_D{D}"] = npy(CostVolume(D, "difference")(L, R))
        out[f"concat_D{D}"] = npy(CostVolume(D, "concat")(L, R))
    # ragged width (W % 4 != 0) and D > W is not exercised by the reference
    # (slicing [:-i] with i >= W yields empty rows) -> keep D <= W.
    torch.manual_seed(102)
    L2 = torch.randn(1, 6, 3, 37)
    R2 = torch.randn(1, 6, 3, 37)
    out["L_ragged"] = npy(L2)
    out["R_ragged"] = npy(R2)
    out["corr_ragged_D9"] = npy(CostVolume(9, "correlation")(L2, R2))
    out["gwc_ragged_D9_G3"] = npy(build_gwc_volume(L2, R2, 9, 3))
    save("g1_volumes", **out)


# ------------------------------------------------------------------ G2
def g2_soft_argmin():
    from models.utils.estimation import DisparityEstimation, DisparityEstimationWithProb

    torch.manual_seed(201)
    vol = 3.0 * torch.randn(2, 12, 5, 40)
    out = {"vol": npy(vol)}
    out["disp_sim"] = npy(DisparityEstimation(12, True)(vol))
    out["disp_cost"] = npy(DisparityEstimation(12, False)(vol))
    out["disp_other_maxdisp"] = npy(DisparityEstimation(48, True)(vol))  # estimation.py:24-28 branch
    d, p = DisparityEstimationWithProb(12, True)(vol)
    out["disp_withprob"] = npy(d)
    out["prob"] = npy(p)
    # large logits (softmax stability)
    big = 40.0 * torch.randn(1, 24, 3, 17)
    out["vol_big"] = npy(big)
    out["disp_big"] = npy(DisparityEstimation(24, True)(big))
    save("g2_soft_argmin", **out)


# ------------------------------------------------------------------ G3
def g3_local_lookup():
    from models.LocalCostVolume.Attempts.fixed_local_cost_volume import (
        LocalCostVolume, build_cost_volume_from_volume)
    from models.LocalCostVolume.Attempts.variance_local_cost_volume import Variance_Based_Local_Cost_Volume

    torch.manual_seed(301)
    D = 24
    vol = 2.0 * torch.randn(2, D, 6, 20)
    disp = torch.rand(2, 1, 6, 20) * (D - 1)
    # hand-placed edge pixels
    disp[0, 0, 0, 0] = 0.0
    disp[0, 0, 0, 1] = 1.5          # disp < r
    disp[0, 0, 0, 2] = D - 1.0      # top integer
    disp[0, 0, 0, 3] = D - 2.25     # disp > D-1-r
    disp[0, 0, 0, 4] = 7.0          # exact integer
    disp[0, 0, 0, 5] = 2.0          # lower bound exactly 0 for r=2
    disp[0, 0, 0, 6] = D - 3.0      # upper bound exactly D-1 for r=2
    disp[0, 0, 0, 7] = 11.5
    out = {"vol": npy(vol), "disp": npy(disp)}
    for (r, S) in ((2, 20), (4, 10), (2.1, 20)):
        for cv in (False, True):
            key = f"fixed_r{r}_S{S}_cv{int(cv)}"
            out[key] = npy(LocalCostVolume(radius=r, sample_points=S)(vol, disp, cv))
    for (g, S) in ((1, 10), (0.5, 20)):
        for cv in (False, True):
            key = f"var_g{g}_S{S}_cv{int(cv)}"
            out[key] = npy(Variance_Based_Local_Cost_Volume(garma=g, sample_points=S)(vol, disp, cv))
    # raw sampler incl. out-of-range / negative candidates
    cand = torch.rand(2, 7, 6, 20) * (D + 6) - 3.0
    cand[0, 0, 0, 0] = 5.0
    cand[0, 1, 0, 0] = -0.0
    cand[0, 2, 0, 0] = D - 1.0
    out["cand"] = npy(cand)
    out["sampled"] = npy(build_cost_volume_from_volume(vol, cand))
    save("g3_local_lookup", **out)


# ------------------------------------------------------------------ G4
def g4_raft_corr():
    from models.IterativeLocalCostVolume.corr.corr import CorrBlock1D, PytorchAlternateCorrBlock1D
    from models.IterativeLocalCostVolume.utils.utils import coords_grid, bilinear_sampler

    torch.manual_seed(401)
    B, C, H, W = 2, 16, 5, 24
    f1 = torch.randn(B, C, H, W)
    f2 = torch.randn(B, C, H, W)
    coords = coords_grid(B, H, W) - torch.cat([torch.rand(B, 1, H, W) * 10.0, torch.zeros(B, 1, H, W)], 1)
    coords[0, 0, 0, 0] = -3.0      # far out of range (left)
    coords[0, 0, 0, 1] = W + 2.0   # far out of range (right)
    coords[0, 0, 0, 2] = 5.0       # exact integer
    coords[0, 0, 0, 3] = W - 1.0
    out = {"f1": npy(f1), "f2": npy(f2), "coords": npy(coords)}
    out["allpairs"] = npy(CorrBlock1D.corr(f1, f2)).reshape(B, H, W, W)
    blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
    for i, lv in enumerate(blk.corr_pyramid):
        out[f"pyr{i}"] = npy(lv).reshape(B * H * W, -1)
    out["lookup_r4_l4"] = npy(blk(coords))
    blk2 = CorrBlock1D(f1, f2, num_levels=2, radius=3)
    out["lookup_r3_l2"] = npy(blk2(coords))
    alt = PytorchAlternateCorrBlock1D(f1, f2, num_levels=4, radius=4)
    out["alt_lookup_r4_l4"] = npy(alt(coords))
    out["coords_grid_3x4"] = npy(coords_grid(2, 3, 4))
    # bilinear_sampler on a 2-D image (general form, mask=True)
    img = torch.randn(2, 3, 6, 9)
    cxy = torch.stack([torch.rand(2, 4, 5) * 11 - 1, torch.rand(2, 4, 5) * 8 - 1], dim=-1)
    s, m = bilinear_sampler(img, cxy, mask=True)
    out["bs_img"], out["bs_coords"], out["bs_out"], out["bs_mask"] = npy(img), npy(cxy), npy(s), npy(m)
    save("g4_raft_corr", **out)


# ------------------------------------------------------------------ G5
def g5_gru_updater():
    from models.LocalCostVolume.Attempts.update import DisparityUpdateDLCWithMask

    torch.manual_seed(501)
    D, h, w = 24, 12, 24
    net = DisparityUpdateDLCWithMask(input_channels=D, hidden_dim=8, sample_points=20).eval()
    fill_state_dict_(net, seed=0.5)
    vol = torch.randn(1, D, h, w)
    disp = torch.rand(1, 1, h, w) * (D - 1)
    left = torch.randn(1, 3, 8 * h, 8 * w)
    right = torch.randn(1, 3, 8 * h, 8 * w)
    hidden = torch.zeros(1, 16, h, w)
    out = {"vol": npy(vol), "disp0": npy(disp), "left": npy(left), "right": npy(right)}
    for it in range(3):
        disp, hidden, mask = net(vol, disp, left, right, hidden, True)
        out[f"disp{it + 1}"] = npy(disp)
        out[f"hidden{it + 1}"] = npy(hidden)
        out[f"mask{it + 1}_sub"] = npy(mask[:, ::37])  # subsample the 576 channels (size)
    # the guidance encoder's pointwise pieces on their own
    from utils.disparity_warper import disp_warp
    p = F.softmax(vol, dim=1)
    cand = torch.arange(D).view(1, D, 1, 1).float()
    d0 = torch.from_numpy(out["disp0"])
    out["uncertainty0"] = npy(p * cand * (cand - d0) * (cand - d0))
    li = F.interpolate(left, size=[h, w], mode="bilinear", align_corners=False)
    ri = F.interpolate(right, size=[h, w], mode="bilinear", align_corners=False)
    out["left_small"], out["right_small"] = npy(li), npy(ri)
    out["error0"] = npy(disp_warp(ri, d0)[0] - li)
    save("g5_gru_updater", **out)


# ------------------------------------------------------------------ G6
def g6_disp_warp():
    from utils.disparity_warper import disp_warp

    torch.manual_seed(601)
    img = torch.randn(2, 3, 7, 19)
    disp = torch.rand(2, 1, 7, 19) * 12.0
    disp[0, 0, 0, 0] = 0.0
    disp[0, 0, 0, 5] = 30.0   # > W
    disp[0, 0, 3, 4] = 4.0    # integer
    out = {"img": npy(img), "disp": npy(disp)}
    for pm in ("border", "zeros"):
        wimg, m = disp_warp(img, disp, padding_mode=pm)
        out[f"warped_{pm}"] = npy(wimg)
        out[f"mask_{pm}"] = npy(m)
    save("g6_disp_warp", **out)


# ------------------------------------------------------------------ G7
def g7_upsample():
    from models.utils.upsample import upsample_convex8, upsample_simple8
    from models.IterativeLocalCostVolume.utils.utils import upflow8

    torch.manual_seed(701)
    disp = torch.rand(2, 1, 5, 7) * 20
    mask8 = torch.randn(2, 576, 5, 7)
    out = {"disp": npy(disp), "mask8": npy(mask8)}
    out["up8"] = npy(upsample_convex8(disp, mask8))
    out["simple8"] = npy(upsample_simple8(disp))
    out["upflow8"] = npy(upflow8(torch.cat([disp, -disp], 1)))

    # RAFT upsample_flow (factor 4, two flow channels): raft_stereo.py:110-122,
    # evaluated through the reference method bound to a stand-in `self`.
    with contextlib.redirect_stdout(io.StringIO()):
        sys.argv = ["x"]
        import types
        t = types.ModuleType("turtle"); t.left = t.right = None; sys.modules["turtle"] = t
        try:
            from models.IterativeLocalCostVolume.raft_stereo import RAFT_Stereo
            ok = True
        except Exception as e:  # pragma: no cover - environment dependent
            print("RAFT_Stereo import failed:", e, file=sys.stderr)
            ok = False
    flow = torch.randn(2, 2, 5, 7)
    mask4 = torch.randn(2, 144, 5, 7)
    out["flow"], out["mask4"] = npy(flow), npy(mask4)
    if ok:
        holder = types.SimpleNamespace(downsample=2)
        out["upflow4"] = npy(RAFT_Stereo.upsample_flow(holder, flow, mask4))
    save("g7_upsample", **out)


# ------------------------------------------------------------------ G8
def g8_raft_update():
    from models.IterativeLocalCostVolume.update.update import BasicMultiUpdateBlock_LZH

    torch.manual_seed(801)
    hd = [16, 16, 16]
    blk = BasicMultiUpdateBlock_LZH(4, 4, 3, 2, hd).eval()
    fill_state_dict_(blk, seed=0.25)
    H, W = 16, 32
    net = [torch.tanh(torch.randn(1, 16, H >> i, W >> i)) for i in range(3)]
    inp = [[0.3 * torch.randn(1, 16, H >> i, W >> i) for _ in range(3)] for i in range(3)]
    corr = torch.randn(1, 36, H, W)
    flow = torch.cat([torch.randn(1, 1, H, W), torch.zeros(1, 1, H, W)], 1)
    out = {"corr": npy(corr), "flow": npy(flow)}
    for i in range(3):
        out[f"net{i}_0"] = npy(net[i])
        for j in range(3):
            out[f"inp{i}{j}"] = npy(inp[i][j])
    net_l = [n.clone() for n in net]
    for it in range(2):
        net_l, mask, delta = blk(net_l, inp, corr, flow, iter32=True, iter16=True)
        for i in range(3):
            out[f"net{i}_{it + 1}"] = npy(net_l[i])
        out[f"mask_{it + 1}"] = npy(mask[:, ::7])
        out[f"delta_{it + 1}"] = npy(delta)
        flow = flow + delta
    save("g8_raft_update", **out)


# ------------------------------------------------------------------ G9
def g9_raft_loop():
    """RAFT_Stereo end to end (4 iterations, test_mode) with RNG-free weights; the tensors the
    (out-of-scope) extractors hand to the hot loop are captured, so the build's re-hosted loop can
    be fed the same inputs and compared with the reference's final outputs."""
    import types
    t = types.ModuleType("turtle"); t.left = t.right = None; sys.modules["turtle"] = t
    sys.argv = ["x"]
    with contextlib.redirect_stdout(io.StringIO()):
        from models.IterativeLocalCostVolume import raft_stereo as RS
    torch.manual_seed(901)
    model = RS.RAFT_Stereo(hidden_dim=[32, 32, 32]).eval()
    fill_state_dict_(model, seed=0.125)
    cap = {}
    orig_init = RS.CorrBlock1D.__init__

    def rec_init(self, fmap1, fmap2, num_levels=4, radius=4):
        cap["fmap1"], cap["fmap2"] = fmap1.clone(), fmap2.clone()
        orig_init(self, fmap1, fmap2, num_levels=num_levels, radius=radius)

    RS.CorrBlock1D.__init__ = rec_init

    def pre_hook(mod, args, kwargs):
        if "net" not in cap:
            cap["net"] = [n.clone() for n in args[0]]
            cap["inp"] = [[c.clone() for c in lvl] for lvl in args[1]]

    hk = model.update_block.register_forward_pre_hook(pre_hook, with_kwargs=True)
    left = torch.rand(1, 3, 64, 96) * 255
    right = torch.rand(1, 3, 64, 96) * 255
    with contextlib.redirect_stdout(io.StringIO()):
        flow_lo, flow_up = model(left, right, iters=4, test_mode=True)
        preds = model(left, right, iters=3, test_mode=False)
    hk.remove()
    RS.CorrBlock1D.__init__ = orig_init
    out = {"fmap1": npy(cap["fmap1"]), "fmap2": npy(cap["fmap2"]), "flow_lo": npy(flow_lo), "flow_up": npy(flow_up)}
    out["keys"] = np.array(sorted(model.state_dict().keys()))   # fill_state_dict_ indexes tensors by this order
    for i in range(3):
        out[f"net{i}"] = npy(cap["net"][i])
        for j in range(3):
            out[f"inp{i}{j}"] = npy(cap["inp"][i][j])
    for k, p in enumerate(preds):
        out[f"pred{k}"] = npy(p)
    save("g9_raft_loop", **out)


def _import_only_standins():
    """scikit-image and torchvision are absent in the build container; the reference modules used here only IMPORT them (never
    execute anything from them on these code paths): empty stand-in modules make the imports succeed (SURVEY 9.3 recipe)."""
    import types
    sk = types.ModuleType("skimage"); sk.io = types.ModuleType("skimage.io"); sk.transform = types.ModuleType("skimage.transform")
    sys.modules.update({"skimage": sk, "skimage.io": sk.io, "skimage.transform": sk.transform})
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional"); tv.transforms = tvt; tvt.functional = tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})


# ------------------------------------------------------------------ G10
def g10_metrics_io():
    """Metrics, disparity rescale and the PFM container."""
    import tempfile
    from losses.squence_loss import EPE_Loss, sequence_loss
    from utils.metric import D1_metric, P1_metric, thres_metric
    from utils.file_io import write_pfm, _read_pfm

    torch.manual_seed(1001)
    gt = torch.rand(2, 1, 9, 13) * 220 - 10        # some <= 0 and some >= 192
    pred = gt + torch.randn(2, 1, 9, 13) * 2.5
    preds = [pred + 0.5 * i for i in range(3)]
    out = {"gt": npy(gt), "pred": npy(pred)}
    out["epe"] = npy(EPE_Loss(pred, gt))
    out["seq"] = npy(sequence_loss(preds, gt))
    out["d1"] = npy(D1_metric(pred, gt))
    out["p1"] = npy(P1_metric(pred, gt))
    out["thres3"] = npy(thres_metric(pred, gt, (gt > 0) & (gt < 192), 3.0))
    # the reference's own scale_disp (dataloader/preprocess.py:257-272; the module imports torchvision / skimage at the top but
    # executes neither on this path: import-only stand-ins, as for G16)
    _import_only_standins()
    from dataloader.preprocess import scale_disp
    d = torch.rand(1, 1, 12, 20) * 50
    out["scale_in"], out["scale_out"] = npy(d), npy(scale_disp(d, (1, 9, 16)))
    img = (torch.rand(7, 11) * 100).numpy().astype(np.float32)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "x.pfm")
        write_pfm(path, img)
        out["pfm_bytes"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        back, scale = _read_pfm(path)
        out["pfm_img"], out["pfm_back"] = img, np.ascontiguousarray(back)
    save("g10_metrics_io", **out)


# ------------------------------------------------------------------ G11
def g11_backward():
    """Gradients of the cfg-2 chain ops from the reference's own autograd (CPU)."""
    from models.utils.build_cost_volume import CostVolume
    from models.CrossAttentionCostVolume.GWC_Cost_Volume import build_gwc_volume
    from models.utils.estimation import DisparityEstimation, DisparityEstimationWithProb
    from models.LocalCostVolume.Attempts.fixed_local_cost_volume import LocalCostVolume

    out = {}
    with torch.enable_grad():
        torch.manual_seed(1101)
        L = torch.randn(2, 8, 4, 40, requires_grad=True)
        R = torch.randn(2, 8, 4, 40, requires_grad=True)
        for (D, G) in ((12, 1), (12, 4), (40, 2)):
            vol = CostVolume(D, "correlation")(L, R) if G == 1 else build_gwc_volume(L, R, D, G)
            cot = torch.randn_like(vol)
            gl, gr = torch.autograd.grad(vol, (L, R), cot)
            out[f"vol_cot_D{D}_G{G}"], out[f"vol_gl_D{D}_G{G}"], out[f"vol_gr_D{D}_G{G}"] = npy(cot), npy(gl), npy(gr)
        out["L"], out["R"] = npy(L), npy(R)
        v = (3.0 * torch.randn(2, 12, 5, 40)).requires_grad_(True)
        cd = torch.randn(2, 5, 40)
        for sim in (True, False):
            (gv,) = torch.autograd.grad(DisparityEstimation(12, sim)(v), v, cd)
            out[f"sa_gvol_sim{int(sim)}"] = npy(gv)
        d, p = DisparityEstimationWithProb(12, True)(v)
        cp = torch.randn_like(p)
        (gv,) = torch.autograd.grad((d, p), v, (cd, cp))
        out["sa_vol"], out["sa_cot_disp"], out["sa_cot_prob"], out["sa_gvol_withprob"] = npy(v), npy(cd), npy(cp), npy(gv)
        vol = (2.0 * torch.randn(2, 24, 6, 20)).requires_grad_(True)
        disp = (torch.rand(2, 1, 6, 20) * 23).requires_grad_(True)
        with torch.no_grad():
            disp[0, 0, 0, 0] = 0.5
            disp[0, 0, 0, 1] = 22.5
            disp[0, 0, 0, 2] = 7.0
        co = torch.randn(2, 1, 6, 20)
        out["ll_vol"], out["ll_disp"], out["ll_cot"] = npy(vol), npy(disp), npy(co)
        for (r, S) in ((2, 20), (4, 10)):
            for cv in (False, True):
                gvv, gdd = torch.autograd.grad(LocalCostVolume(r, S)(vol, disp, cv), (vol, disp), co)
                out[f"ll_gvol_r{r}_S{S}_cv{int(cv)}"], out[f"ll_gdisp_r{r}_S{S}_cv{int(cv)}"] = npy(gvv), npy(gdd)
        # convex upsampling (A13): f=8 single channel and the RAFT flavour f=4 with two flow channels
        from models.utils.upsample import upsample_convex8
        d8 = (torch.rand(1, 1, 3, 70) * 20).requires_grad_(True)
        m8 = torch.randn(1, 576, 3, 70, requires_grad=True)
        up = upsample_convex8(d8, m8)
        cu = torch.randn_like(up)
        g_d8, g_m8 = torch.autograd.grad(up, (d8, m8), cu)
        out["up8_disp"], out["up8_mask"], out["up8_cot"] = npy(d8), npy(m8), npy(cu)
        out["up8_gdisp"], out["up8_gmask_sub"] = npy(g_d8), npy(g_m8[:, ::31])
        import types
        t = types.ModuleType("turtle"); t.left = t.right = None; sys.modules["turtle"] = t
        sys.argv = ["x"]
        with contextlib.redirect_stdout(io.StringIO()):
            from models.IterativeLocalCostVolume.raft_stereo import RAFT_Stereo
        fl = torch.randn(1, 2, 6, 9, requires_grad=True)
        m4 = torch.randn(1, 144, 6, 9, requires_grad=True)
        up4 = RAFT_Stereo.upsample_flow(types.SimpleNamespace(downsample=2), fl, m4)
        c4 = torch.randn_like(up4)
        g_fl, g_m4 = torch.autograd.grad(up4, (fl, m4), c4)
        out["up4_flow"], out["up4_mask"], out["up4_cot"] = npy(fl), npy(m4), npy(c4)
        out["up4_gflow"], out["up4_gmask"] = npy(g_fl), npy(g_m4)
        # A11 module: two iterations, gradients w.r.t. inputs and every parameter
        from models.LocalCostVolume.Attempts.update import DisparityUpdateDLCWithMask
        Dm, hh, ww = 24, 8, 16
        net = DisparityUpdateDLCWithMask(input_channels=Dm, hidden_dim=8, sample_points=20).eval()
        fill_state_dict_(net, seed=0.5)
        volm = torch.randn(1, Dm, hh, ww, requires_grad=True)
        d0 = (torch.rand(1, 1, hh, ww) * (Dm - 1)).requires_grad_(True)
        lim, rim = torch.randn(1, 3, 8 * hh, 8 * ww), torch.randn(1, 3, 8 * hh, 8 * ww)
        hid = torch.zeros(1, 16, hh, ww)
        dd, hidn, mk = net(volm, d0, lim, rim, hid, True)
        dd2, hidn2, mk2 = net(volm, dd, lim, rim, hidn, True)
        cdd, cmk, chid = torch.randn_like(dd2), torch.randn_like(mk2), torch.randn_like(hidn2)
        loss = (dd2 * cdd).sum() + (mk2 * cmk).sum() * 0.01 + (hidn2 * chid).sum() + (dd * cdd).sum()
        names = [n for n, _ in net.named_parameters()]
        grads = torch.autograd.grad(loss, [volm, d0] + [p for _, p in net.named_parameters()])
        out["dlc_vol"], out["dlc_disp0"], out["dlc_left"], out["dlc_right"] = npy(volm), npy(d0), npy(lim), npy(rim)
        out["dlc_cdd"], out["dlc_cmk_sub"], out["dlc_chid"] = npy(cdd), npy(cmk[:, ::37]), npy(chid)
        out["dlc_cmk_seed"] = np.array([0])
        out["dlc_gvol"], out["dlc_gdisp0"] = npy(grads[0]), npy(grads[1])
        out["dlc_param_names"] = np.array(names)
        for n_, g_ in zip(names, grads[2:]):
            out["dlc_gp_" + n_] = npy(g_) if g_.numel() < 20000 else npy(g_.flatten()[::17])
        out["dlc_cmk_full"] = npy(cmk)
    save("g11_backward", **out)


# ------------------------------------------------------------------ G12
def g12_raft_backward():
    """Gradients of the RAFT_Stereo hot loop from the reference's autograd: 3 iterations, loss over all
    predictions; cotangents w.r.t. the tensors the extractors hand over and the update block's parameters."""
    import types
    t = types.ModuleType("turtle"); t.left = t.right = None; sys.modules["turtle"] = t
    sys.argv = ["x"]
    with contextlib.redirect_stdout(io.StringIO()):
        from models.IterativeLocalCostVolume import raft_stereo as RS
    torch.manual_seed(1201)
    with torch.enable_grad():
        model = RS.RAFT_Stereo(hidden_dim=[32, 32, 32]).eval()
        fill_state_dict_(model, seed=0.125)
        cap = {}
        orig_init = RS.CorrBlock1D.__init__

        def rec_init(self, fmap1, fmap2, num_levels=4, radius=4):
            fmap1.retain_grad(); fmap2.retain_grad()
            cap["fmap1"], cap["fmap2"] = fmap1, fmap2
            orig_init(self, fmap1, fmap2, num_levels=num_levels, radius=radius)

        RS.CorrBlock1D.__init__ = rec_init

        def pre_hook(mod, args, kwargs):
            if "net" not in cap:
                for x in args[0]:
                    x.retain_grad()
                for lvl in args[1]:
                    for c in lvl:
                        c.retain_grad()
                cap["net"] = list(args[0])
                cap["inp"] = [list(lvl) for lvl in args[1]]

        hk = model.update_block.register_forward_pre_hook(pre_hook, with_kwargs=True)
        left = torch.rand(1, 3, 64, 96) * 255
        right = torch.rand(1, 3, 64, 96) * 255
        with contextlib.redirect_stdout(io.StringIO()):
            preds = model(left, right, iters=3, test_mode=False)
        hk.remove()
        RS.CorrBlock1D.__init__ = orig_init
        cots = [torch.randn_like(p) for p in preds]
        loss = sum((p * c).sum() for p, c in zip(preds, cots))
        loss.backward()
    out = {"fmap1": npy(cap["fmap1"]), "fmap2": npy(cap["fmap2"]),
           "g_fmap1": npy(cap["fmap1"].grad), "g_fmap2": npy(cap["fmap2"].grad)}
    out["keys"] = np.array(sorted(model.state_dict().keys()))
    for i in range(3):
        out[f"net{i}"], out[f"g_net{i}"] = npy(cap["net"][i]), npy(cap["net"][i].grad)
        for j in range(3):
            out[f"inp{i}{j}"], out[f"g_inp{i}{j}"] = npy(cap["inp"][i][j]), npy(cap["inp"][i][j].grad)
    for k, c in enumerate(cots):
        out[f"cot{k}"] = npy(c)
    names = [n for n, _ in model.update_block.named_parameters()]
    out["param_names"] = np.array(names)
    for n_, p_ in model.update_block.named_parameters():
        g_ = p_.grad
        out["gp_" + n_] = npy(g_) if g_.numel() < 20000 else npy(g_.flatten()[::37])
    save("g12_raft_backward", **out)


# ------------------------------------------------------------------ cfg-1 check
# ------------------------------------------------------------------ G13
def g13_flow_init_padder():
    """(a) RAFT_Stereo started from a given disparity field (`flow_init`, raft_stereo.py:163-164), 3 iterations, test_mode;
    the tensors the extractors hand to the hot loop are captured as in G9.  (b) InputPadder.pad / .unpad VALUES
    (utils/utils.py:7-26), both modes, divis_by 8 and 32."""
    import types
    t = types.ModuleType("turtle"); t.left = t.right = None; sys.modules["turtle"] = t
    sys.argv = ["x"]
    with contextlib.redirect_stdout(io.StringIO()):
        from models.IterativeLocalCostVolume import raft_stereo as RS
    from models.IterativeLocalCostVolume.utils.utils import InputPadder
    torch.manual_seed(1301)
    model = RS.RAFT_Stereo(hidden_dim=[32, 32, 32]).eval()
    fill_state_dict_(model, seed=0.125)
    cap = {}
    orig_init = RS.CorrBlock1D.__init__

    def rec_init(self, fmap1, fmap2, num_levels=4, radius=4):
        cap["fmap1"], cap["fmap2"] = fmap1.clone(), fmap2.clone()
        orig_init(self, fmap1, fmap2, num_levels=num_levels, radius=radius)

    RS.CorrBlock1D.__init__ = rec_init

    def pre_hook(mod, args, kwargs):
        if "net" not in cap:
            cap["net"] = [n.clone() for n in args[0]]
            cap["inp"] = [[c.clone() for c in lvl] for lvl in args[1]]

    hk = model.update_block.register_forward_pre_hook(pre_hook, with_kwargs=True)
    left = torch.rand(1, 3, 64, 96) * 255
    right = torch.rand(1, 3, 64, 96) * 255
    flow_init = torch.zeros(1, 2, 16, 24)
    flow_init[:, 0] = -6.0 * torch.rand(1, 16, 24)        # negative x-flow = positive disparity
    with contextlib.redirect_stdout(io.StringIO()):
        flow_lo, flow_up = model(left, right, iters=3, flow_init=flow_init, test_mode=True)
    hk.remove()
    RS.CorrBlock1D.__init__ = orig_init
    out = {"fmap1": npy(cap["fmap1"]), "fmap2": npy(cap["fmap2"]), "flow_init": npy(flow_init), "flow_lo": npy(flow_lo),
           "flow_up": npy(flow_up)}
    for i in range(3):
        out[f"net{i}"] = npy(cap["net"][i])
        for j in range(3):
            out[f"inp{i}{j}"] = npy(cap["inp"][i][j])
    x = torch.randn(2, 3, 37, 50)
    out["pad_x"] = npy(x)
    for mode in ("sintel", "kitti"):
        for div in (8, 32):
            p = InputPadder(x.shape, mode=mode, divis_by=div)
            (xp,) = p.pad(x)
            out[f"pad_{mode}_{div}"] = npy(xp)
            out[f"unpad_{mode}_{div}"] = npy(p.unpad(xp))
    p = InputPadder((1, 3, 540, 960), divis_by=32)          # cfg-4: 540 -> 544
    (xp,) = p.pad(x[:1, :, :4, :8].repeat(1, 1, 135, 120))
    out["pad_540x960_rows"] = npy(xp[0, 0, [0, 1, 2, 541, 542, 543], :16])
    save("g13_flow_init_padder", **out)


# ------------------------------------------------------------------ G14
def g14_extractor_epilogue():
    """InstanceNorm2d + ReLU as the reference's extractor applies them (extractor.py:135,187-188 and the residual blocks'
    norm1/relu :38-41,63): the tensors entering norm1 / leaving relu1 of BasicEncoder(norm_fn='instance') and of its first
    ResidualBlock are captured with hooks; plus an affine InstanceNorm2d (weight / bias) without ReLU."""
    from models.IterativeLocalCostVolume.extractor.extractor import BasicEncoder
    torch.manual_seed(1401)
    enc = BasicEncoder(output_dim=32, norm_fn="instance", downsample=2).eval()
    cap = {}
    enc.norm1.register_forward_hook(lambda m, a, o: cap.__setitem__("x1", a[0].clone()))
    enc.relu1.register_forward_hook(lambda m, a, o: cap.__setitem__("y1", o.clone()))
    blk = enc.layer1[0]
    blk.norm1.register_forward_hook(lambda m, a, o: cap.__setitem__("x2", a[0].clone()))
    img = torch.rand(2, 3, 20, 36) * 2 - 1
    enc(img)
    out = {"x1": npy(cap["x1"]), "y1": npy(cap["y1"]), "x2": npy(cap["x2"]),
           "y2": npy(torch.relu(blk.norm1(cap["x2"])))}
    aff = torch.nn.InstanceNorm2d(6, eps=1e-3, affine=True).eval()
    aff.weight.data = torch.linspace(0.5, 2.0, 6)
    aff.bias.data = torch.linspace(-1.0, 1.0, 6)
    xa = torch.randn(1, 6, 9, 12) * 3 + 5           # large mean relative to the spread
    out.update(xa=npy(xa), ya=npy(aff(xa)), wa=npy(aff.weight.data), ba=npy(aff.bias.data))
    save("g14_extractor_epilogue", **out)


# ------------------------------------------------------------------ G15
def g15_backward2():
    """Gradients from the reference's own autograd for the ops that became differentiable in round 2: A6
    (Variance_Based_Local_Cost_Volume, both validity modes) and A2 ('difference' / 'concat' volumes)."""
    from models.utils.build_cost_volume import CostVolume
    from models.LocalCostVolume.Attempts.variance_local_cost_volume import Variance_Based_Local_Cost_Volume

    out = {}
    with torch.enable_grad():
        torch.manual_seed(1501)
        vol = (2.0 * torch.randn(2, 24, 6, 28)).requires_grad_(True)
        disp = (torch.rand(2, 1, 6, 28) * 23).requires_grad_(True)
        with torch.no_grad():
            disp[0, 0, 0, 0] = 0.5
            disp[0, 0, 0, 1] = 22.5
            disp[0, 0, 0, 2] = 7.0
            vol[0, :, 1, 5] = 0.0
            vol[0, 9, 1, 5] = 12.0          # a sharply peaked column: small sigma
            disp[0, 0, 1, 5] = 9.25
        co = torch.randn(2, 1, 6, 28)
        out["vl_vol"], out["vl_disp"], out["vl_cot"] = npy(vol), npy(disp), npy(co)
        for (gm, S) in ((1.0, 10), (2.0, 20), (0.5, 7)):
            for cv in (False, True):
                o = Variance_Based_Local_Cost_Volume(gm, S)(vol, disp, cv)
                gvv, gdd = torch.autograd.grad(o, (vol, disp), co)
                out[f"vl_out_g{gm}_S{S}_cv{int(cv)}"] = npy(o)
                out[f"vl_gvol_g{gm}_S{S}_cv{int(cv)}"], out[f"vl_gdisp_g{gm}_S{S}_cv{int(cv)}"] = npy(gvv), npy(gdd)
        L = torch.randn(2, 3, 4, 20, requires_grad=True)
        R = torch.randn(2, 3, 4, 20, requires_grad=True)
        out["sv_L"], out["sv_R"] = npy(L), npy(R)
        for kind in ("difference", "concat"):
            for D in (1, 7, 20, 26):          # D == W and D > W included
                v = CostVolume(D, kind)(L, R)
                cot = torch.randn_like(v)
                gl, gr = torch.autograd.grad(v, (L, R), cot)
                out[f"sv_cot_{kind}_D{D}"], out[f"sv_gl_{kind}_D{D}"], out[f"sv_gr_{kind}_D{D}"] = npy(cot), npy(gl), npy(gr)
    save("g15_backward2", **out)


# ------------------------------------------------------------------ G16
def g16_dataset():
    """StereoDataset + the geometric transforms (dataloader/SceneflowLoader.py, dataloader/transforms.py, utils/file_io.py) on a
    tiny synthetic tree written to a temp dir: two SceneFlow-style samples (PNG + PFM) in 'train' mode through
    Compose([RandomCrop(validate=True), ToTensor(), Normalize(...)]), and a KITTI-style 16-bit PNG disparity.  scikit-image and torchvision are absent here: they are only imported (never executed) by the
    code paths used, so import-only stand-ins are installed first (SURVEY 9.3 recipe).  The fixture keeps the source arrays,
    so the test can re-create the files."""
    import tempfile
    import types
    from PIL import Image
    _import_only_standins()
    from dataloader.SceneflowLoader import StereoDataset
    from dataloader import transforms as T
    from utils.file_io import write_pfm, read_disp

    rng = np.random.default_rng(1601)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        lines = []
        for i in range(2):
            os.makedirs(os.path.join(td, "frames", f"{i}"), exist_ok=True)
            l = rng.integers(0, 256, (20, 36, 3), dtype=np.uint8)
            r = rng.integers(0, 256, (20, 36, 3), dtype=np.uint8)
            d = (rng.random((20, 36)) * 30).astype(np.float32)
            Image.fromarray(l).save(os.path.join(td, "frames", f"{i}", "left.png"))
            Image.fromarray(r).save(os.path.join(td, "frames", f"{i}", "right.png"))
            write_pfm(os.path.join(td, "frames", f"{i}", "disp.pfm"), d)
            lines.append(f"frames/{i}/left.png frames/{i}/right.png frames/{i}/disp.pfm")
            out[f"src_left{i}"], out[f"src_right{i}"], out[f"src_disp{i}"] = l, r, d
        lst = os.path.join(td, "train.list")
        open(lst, "w").write("\n".join(lines) + "\n")
        mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
        ds = StereoDataset(td, lst, lst, mode="train", save_filename=True,
                           transform=T.Compose([T.RandomCrop(16, 32, validate=True), T.ToTensor(), T.Normalize(mean, std)]))
        for i in range(2):
            smp = ds[i]
            out[f"crop_left{i}"], out[f"crop_right{i}"], out[f"crop_disp{i}"] = npy(smp["img_left"]), npy(smp["img_right"]), npy(smp["gt_disp"])
            out[f"name{i}"] = np.array([smp["left_name"]])
        # (RandomCrop's zero-padding branch calls np.lib.pad, which numpy 2 removed: the reference raises there on this
        # interpreter, so that branch is restated from the source text and checked against np.pad in the test, unpinned.)
        k = (rng.random((6, 9)) * 200 * 256).astype(np.uint16)
        Image.fromarray(k).save(os.path.join(td, "kitti_disp.png"))
        out["kitti_png"], out["kitti_disp"] = k, read_disp(os.path.join(td, "kitti_disp.png")).astype(np.float32)
    save("g16_dataset", **out)


def check_oracle_cfg1():
    """Not a fixture: proves the C oracle == the reference at BASELINE
    configs[0] (1 pair 384x768 D=64: corr -> soft-argmin -> local lookup)."""
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
    import oracle
    from models.utils.build_cost_volume import CostVolume
    from models.utils.estimation import DisparityEstimation
    from models.LocalCostVolume.Attempts.fixed_local_cost_volume import LocalCostVolume

    torch.manual_seed(0)
    L = torch.randn(1, 32, 384, 768)
    R = torch.randn(1, 32, 384, 768)
    vol = CostVolume(64, "correlation")(L, R)
    disp = DisparityEstimation(64)(vol)
    d2 = LocalCostVolume(2, 20)(vol, disp.unsqueeze(1), True)
    ovol = oracle.corr_volume(npy(L), npy(R), 64)
    odisp = oracle.soft_argmin(ovol)
    od2 = oracle.local_lookup(ovol, odisp[:, None], 2.0, 20, True)
    print("cfg-1 oracle vs reference: vol %.3g disp %.3g lookup %.3g" % (
        np.abs(ovol - npy(vol)).max(), np.abs(odisp - npy(disp)).max(), np.abs(od2 - npy(d2)).max()))


if __name__ == "__main__":
    CHECK = "--check" in sys.argv
    if "--only-g10" in sys.argv:
        g10_metrics_io()
        sys.exit(0)
    g1_volumes()
    g2_soft_argmin()
    g3_local_lookup()
    g4_raft_corr()
    g5_gru_updater()
    g6_disp_warp()
    g7_upsample()
    g8_raft_update()
    g9_raft_loop()
    g10_metrics_io()
    g11_backward()
    g12_raft_backward()
    g13_flow_init_padder()
    g14_extractor_epilogue()
    g15_backward2()
    g16_dataset()
    if CHECK:
        check_oracle_cfg1()
