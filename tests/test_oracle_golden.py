"""CPU: the C oracle (oracle/stereo_oracle.c) against golden tensors produced by
the real reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

TOL = dict(rtol=1e-5, atol=2e-6)


def close(a, b, **kw):
    t = dict(TOL)
    t.update(kw)
    np.testing.assert_allclose(a, b, **t)


def test_volumes_g1(oracle, golden):
    g = golden("g1_volumes")
    L, R = g["L"], g["R"]
    for D in (1, 12, 40):
        close(oracle.corr_volume(L, R, D), g[f"corr_D{D}"])
    for (D, G) in ((1, 4), (12, 1), (12, 4), (12, 16), (40, 2)):
        close(oracle.gwc_volume(L, R, D, G), g[f"gwc_D{D}_G{G}"])
    for D in (1, 12):
        close(oracle.diff_volume(L, R, D), g[f"diff_D{D}"], atol=0, rtol=0)
        close(oracle.concat_volume(L, R, D), g[f"concat_D{D}"], atol=0, rtol=0)
    close(oracle.corr_volume(g["L_ragged"], g["R_ragged"], 9), g["corr_ragged_D9"])
    close(oracle.gwc_volume(g["L_ragged"], g["R_ragged"], 9, 3), g["gwc_ragged_D9_G3"])
    with pytest.raises(AssertionError):
        oracle.gwc_volume(L, R, 4, 5)


def test_identities_i1_i4(oracle, golden):
    g = golden("g1_volumes")
    L, R = g["L"], g["R"]
    corr = oracle.corr_volume(L, R, 12)
    close(oracle.group_mean(oracle.gwc_volume(L, R, 12, 4)), corr, atol=1e-6)  # I1
    close(oracle.gwc_volume(L, R, 12, 1)[:, 0], corr, atol=1e-6)               # I4


def test_soft_argmin_g2(oracle, golden):
    g = golden("g2_soft_argmin")
    close(oracle.soft_argmin(g["vol"], True), g["disp_sim"], atol=2e-5)
    close(oracle.soft_argmin(g["vol"], False), g["disp_cost"], atol=2e-5)
    close(oracle.soft_argmin(g["vol"], True), g["disp_other_maxdisp"], atol=2e-5)
    d, p = oracle.soft_argmin(g["vol"], True, with_prob=True)
    close(d, g["disp_withprob"], atol=2e-5)
    close(p, g["prob"], atol=1e-6)
    close(oracle.soft_argmin(g["vol_big"], True), g["disp_big"], atol=2e-5)


def test_local_lookup_g3(oracle, golden):
    g = golden("g3_local_lookup")
    vol, disp = g["vol"], g["disp"]
    for (r, S) in ((2, 20), (4, 10), (2.1, 20)):
        for cv in (False, True):
            close(oracle.local_lookup(vol, disp, r, S, cv), g[f"fixed_r{r}_S{S}_cv{int(cv)}"], atol=2e-5)
    for (gm, S) in ((1, 10), (0.5, 20)):
        for cv in (False, True):
            close(oracle.variance_lookup(vol, disp, gm, S, cv), g[f"var_g{gm}_S{S}_cv{int(cv)}"], atol=5e-5)
    close(oracle.sample_volume(vol, g["cand"]), g["sampled"], atol=1e-5)


def test_identity_i5(oracle, golden):
    g = golden("g3_local_lookup")
    vol, disp = g["vol"], g["disp"]
    D = vol.shape[1]
    out = oracle.local_lookup(vol, disp, 2.0, 20, True)
    invalid = (disp - 2.0 < 0) | (disp + 2.0 >= D - 1)
    assert invalid.any() and (~invalid).any()
    close(out[invalid], disp[invalid], atol=1e-5)


def test_raft_corr_g4(oracle, golden):
    g = golden("g4_raft_corr")
    f1, f2, coords = g["f1"], g["f2"], g["coords"]
    ap = oracle.allpairs_corr(f1, f2)
    close(ap, g["allpairs"], atol=1e-5)
    pyr = oracle.corr_pyramid(ap, 4)
    for i, lv in enumerate(pyr):
        close(lv, g[f"pyr{i}"], atol=1e-5)
    close(oracle.corr_lookup(pyr, coords, 4, 4), g["lookup_r4_l4"], atol=2e-5)
    close(oracle.corr_lookup(oracle.corr_pyramid(ap, 2), coords, 3, 2), g["lookup_r3_l2"], atol=2e-5)
    close(oracle.alt_corr_lookup(f1, f2, coords, 4, 4), g["alt_lookup_r4_l4"], atol=5e-5)


def test_identity_i2(oracle, golden):
    """I2: corr volume band == all-pairs band up to the scale (C vs sqrt(C))."""
    g = golden("g4_raft_corr")
    f1, f2 = g["f1"], g["f2"]
    B, C, H, W = f1.shape
    ap = oracle.allpairs_corr(f1, f2)
    vol = oracle.corr_volume(f1, f2, 8)
    for d in range(8):
        for w in range(d, W):
            close(vol[:, d, :, w] * C, ap[:, :, w, w - d] * np.sqrt(C), atol=2e-5)


def test_disp_warp_g6(oracle, golden):
    g = golden("g6_disp_warp")
    for pm in ("border", "zeros"):
        w, m = oracle.disp_warp(g["img"], g["disp"], pm)
        close(w, g[f"warped_{pm}"], atol=1e-5)
        # the 0.9999 threshold can flip on 1-ulp differences: allow a handful
        assert (m != g[f"mask_{pm}"]).mean() < 0.01


def test_upsample_g7(oracle, golden):
    g = golden("g7_upsample")
    close(oracle.upsample_convex(g["disp"], g["mask8"], 8), g["up8"], atol=2e-5)
    close(oracle.upsample_convex(g["flow"], g["mask4"], 4), g["upflow4"], atol=2e-5)
    close(oracle.upsample_simple(g["disp"], 8), g["simple8"], atol=2e-5)
    # I8: all-zero mask == zero-padded 3x3 box filter of f*disp, replicated f x f
    d = g["disp"]
    up = oracle.upsample_convex(d, np.zeros_like(g["mask8"]), 8)
    pad = np.pad(8 * d, ((0, 0), (0, 0), (1, 1), (1, 1)))
    box = sum(pad[:, :, i:i + d.shape[2], j:j + d.shape[3]] for i in range(3) for j in range(3)) / 9
    close(up, np.repeat(np.repeat(box, 8, 2), 8, 3), atol=2e-5)


def test_gru_updater_pieces_g5(oracle, golden):
    g = golden("g5_gru_updater")
    close(oracle.uncertainty_volume(g["vol"], g["disp0"]), g["uncertainty0"], atol=2e-4, rtol=1e-4)
    h, w = g["disp0"].shape[2:]
    close(oracle.resize_bilinear(g["left"], (h, w), False), g["left_small"], atol=1e-5)
    wr, _ = oracle.disp_warp(g["right_small"], g["disp0"])
    close(wr - g["left_small"], g["error0"], atol=2e-5)


def test_cross_attention_identity_i3(oracle, golden):
    """A7 is parity-unpinned by the reference (empty file); identity I3 pins it
    to A1 + A4 and to a gather of the value features."""
    g = golden("g1_volumes")
    L, R = g["L"], g["R"]
    D = 12
    out, disp = oracle.cross_attention(L, R, None, D, mask="zero", scale="mean")
    vol = oracle.corr_volume(L, R, D)
    dref, p = oracle.soft_argmin(vol, True, with_prob=True)
    close(disp, dref, atol=2e-5)
    B, C, H, W = L.shape
    ref = np.zeros_like(R)
    for d in range(D):
        ref[:, :, :, d:] += p[:, d:d + 1, :, d:] * R[:, :, :, :W - d]
    close(out, ref, atol=2e-5)
    # neginf mask: probability mass only on in-image taps -> disp[w] <= w
    _, dn = oracle.cross_attention(L, R, None, D, mask="neginf", scale="sqrt_c")
    assert (dn <= np.arange(W)[None, None, :] + 1e-4).all()


def test_instance_norm_relu_golden(oracle, golden):
    """N2 extractor epilogue: the oracle against tensors captured inside the reference's BasicEncoder(norm_fn='instance')
    (norm1 -> relu1, a residual block's norm + relu) and an affine InstanceNorm2d without ReLU."""
    g = golden("g14_extractor_epilogue")
    np.testing.assert_allclose(oracle.instance_norm_relu(g["x1"]), g["y1"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(oracle.instance_norm_relu(g["x2"]), g["y2"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(oracle.instance_norm_relu(g["xa"], g["wa"], g["ba"], eps=1e-3, relu=False), g["ya"],
                               rtol=1e-5, atol=5e-6)
