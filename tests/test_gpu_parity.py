"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on seeded inputs,
against the golden fixtures generated from the real reference, and -- at BASELINE.json's
full sizes -- through size-independent identities.

Tolerances (fp32 path; BASELINE north_star: disparity within 1e-3 of the reference):
  volumes / features : 2e-5 abs
  disparities        : max-abs 1e-3 (the north-star bar) AND mean-abs 2.5e-4; measured max-abs is
                       ~3e-4 at D=192 on flat (random-feature) softmaxes, i.e. 3e-6 relative --
                       fp32 summation-order noise of a 192-term expectation around 96.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VOL_TOL = dict(rtol=1e-5, atol=2e-5)
DISP_TOL = dict(rtol=0, atol=1e-3, mean_atol=2.5e-4)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def close(a, b, **tol):
    a = host(a) if isinstance(a, torch.Tensor) else a
    mean_tol = tol.pop("mean_atol", None)
    np.testing.assert_allclose(a, b, **tol)
    if mean_tol is not None and a.size:
        assert float(np.abs(a - b).mean()) <= mean_tol, float(np.abs(a - b).mean())


@pytest.fixture(scope="module")
def S():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    import stereoformer_amd as s
    s._native.load()
    return s


# ------------------------------------------------------------------ volumes (A1/A2/A3)
def test_volumes_golden(S, golden):
    g = golden("g1_volumes")
    L, R = dev(g["L"]), dev(g["R"])
    for D in (1, 12, 40):
        close(S.CostVolume(D, "correlation")(L, R), g[f"corr_D{D}"], **VOL_TOL)
    for (D, G) in ((1, 4), (12, 1), (12, 4), (12, 16), (40, 2)):
        close(S.build_gwc_volume(L, R, D, G), g[f"gwc_D{D}_G{G}"], **VOL_TOL)
    for D in (1, 12):
        close(S.CostVolume(D, "difference")(L, R), g[f"diff_D{D}"], rtol=0, atol=0)
        close(S.CostVolume(D, "concat")(L, R), g[f"concat_D{D}"], rtol=0, atol=0)
        close(S.build_concat_volume(L, R, D), g[f"concat_D{D}"], rtol=0, atol=0)
    Lr, Rr = dev(g["L_ragged"]), dev(g["R_ragged"])
    close(S.CostVolume(9, "correlation")(Lr, Rr), g["corr_ragged_D9"], **VOL_TOL)
    close(S.build_gwc_volume(Lr, Rr, 9, 3), g["gwc_ragged_D9_G3"], **VOL_TOL)
    close(S.groupwise_correlation(L, R, 4), g["gwc_D1_G4"][:, :, 0], **VOL_TOL)


@pytest.mark.parametrize("shape", [
    (2, 32, 7, 300, 64, 8),     # W spans two tiles, partial second tile
    (1, 32, 3, 960, 192, 8),    # cfg-2 row geometry
    (1, 256, 4, 96, 24, 1),     # reference-native: 256 channels at 1/8 scale, D=24
    (1, 6, 5, 37, 9, 3),        # ragged W (scalar path), channels/group = 2
    (2, 12, 3, 520, 200, 4),    # D not a multiple of the chunk, multi-tile
    (1, 8, 2, 16, 16, 2),       # D == W
    (1, 5, 2, 33, 130, 5),      # D > W, channels/group = 1, two passes (D > 8*16)
])
def test_volumes_vs_oracle(S, oracle, shape):
    B, C, H, W, D, G = shape
    rng = np.random.default_rng(hash(shape) % 2**32)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32)
    R = rng.standard_normal((B, C, H, W), dtype=np.float32)
    close(S.CostVolume(D, "correlation")(dev(L), dev(R)), oracle.corr_volume(L, R, D), **VOL_TOL)
    close(S.build_gwc_volume(dev(L), dev(R), D, G), oracle.gwc_volume(L, R, D, G), **VOL_TOL)


@pytest.mark.parametrize("shape", [(2, 3, 4, 40, 12), (1, 2, 3, 37, 9), (1, 2, 2, 1028, 21), (1, 1, 2, 64, 100),
                                   (1, 2, 2, 960, 192)])
def test_shift_volumes_vs_oracle(S, oracle, shape):
    """A2 difference / concat volumes, bit-exact (copies and one subtraction): row-contiguous fast path
    (W % 4 == 0, incl. W > 1024 and D > W) and the scalar path (ragged W)."""
    B, C, H, W, D = shape
    rng = np.random.default_rng(41)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32)
    R = rng.standard_normal((B, C, H, W), dtype=np.float32)
    close(S.CostVolume(D, "difference")(dev(L), dev(R)), oracle.diff_volume(L, R, D), rtol=0, atol=0)
    close(S.CostVolume(D, "concat")(dev(L), dev(R)), oracle.concat_volume(L, R, D), rtol=0, atol=0)


def test_volume_valu_store_kernel_shapes(S, oracle):
    """Store-only group-wise build on the VALU pipeline kernel (the fallback of the matrix-core group pass) against the
    oracle: ragged last tile columns, W > 1024 (two workgroups per row), D not a multiple of the disparities a workgroup
    covers, channels/group in {1, 2, 4}."""
    rng = np.random.default_rng(17)
    with S._native.test_options(corr_mfma=0):
        for (B, C, H, W, D, G) in ((1, 32, 5, 960, 192, 8), (2, 8, 6, 1300, 100, 4), (1, 6, 9, 520, 97, 6), (1, 8, 3, 260, 40, 2)):
            L = rng.standard_normal((B, C, H, W), dtype=np.float32)
            R = rng.standard_normal((B, C, H, W), dtype=np.float32)
            close(S.build_gwc_volume(dev(L), dev(R), D, G), oracle.gwc_volume(L, R, D, G), **VOL_TOL)


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("shape", [(1, 32, 5, 960, 192), (2, 12, 6, 520, 150), (1, 6, 9, 260, 97), (2, 32, 3, 300, 64),
                                   (1, 5, 2, 36, 7), (1, 8, 2, 16, 16), (1, 32, 2, 1300, 33), (1, 3, 4, 8, 1),
                                   (1, 16, 3, 1248, 256), (1, 8, 2, 400, 230), (1, 8, 2, 64, 100), (3, 4, 1, 4, 3)])
def test_corr_volume_mfma_vs_valu_vs_oracle(S, oracle, shape, mfma):
    """A1 and A1+A4 (volume stored and regressed in one pass) on the f32 matrix cores (C <= 32, D <= 192, W % 4 == 0;
    LDS skew sheet, persistent workgroups) and on the VALU kernels: both against the oracle -- every band-size class,
    partial last tiles, D == W, both softmax signs, several tiles per persistent workgroup."""
    B, C, H, W, D = shape
    rng = np.random.default_rng(43)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32) * 2
    R = rng.standard_normal((B, C, H, W), dtype=np.float32) * 2
    ref = oracle.corr_volume(L, R, D)
    with S._native.test_options(corr_mfma=mfma):
        close(S.CostVolume(D)(dev(L), dev(R)), ref, **VOL_TOL)
        for sim in (True, False):
            v, d = S.corr_volume_with_disparity(dev(L), dev(R), D, match_similarity=sim)
            close(v, ref, **VOL_TOL)
            close(d, oracle.soft_argmin(ref, sim), **DISP_TOL)
        with S._native.test_options(tiles_per_wg=3):
            v, d = S.corr_volume_with_disparity(dev(L), dev(R), D)
        close(v, ref, **VOL_TOL)
        close(d, oracle.soft_argmin(ref), **DISP_TOL)


@pytest.mark.parametrize("mode", ["fused", "groups_mfma_mean_separate", "valu"])
@pytest.mark.parametrize("shape", [(1, 32, 5, 960, 192, 8), (2, 12, 6, 520, 150, 3), (1, 32, 3, 300, 64, 2), (1, 8, 4, 260, 97, 4),
                                   (2, 16, 2, 36, 7, 1), (1, 24, 2, 1300, 40, 12), (1, 8, 2, 64, 100, 2), (2, 32, 1, 8, 5, 16)])
def test_gwc_volume_group_pass_mfma(S, oracle, shape, mode):
    """A3 alone and A3 + group mean + A4 in ONE kernel (group pass on the matrix cores, channels per group in
    {2, 4, 8, 16}) against the oracle, next to the two-kernel and the VALU paths."""
    B, C, H, W, D, G = shape
    rng = np.random.default_rng(47)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32) * 2
    R = rng.standard_normal((B, C, H, W), dtype=np.float32) * 2
    ovol = oracle.gwc_volume(L, R, D, G)
    omean = oracle.group_mean(ovol)
    with S._native.test_options(corr_mfma=0 if mode == "valu" else 1,
                                gwc_fused=0 if mode == "groups_mfma_mean_separate" else 1):
        close(S.build_gwc_volume(dev(L), dev(R), D, G), ovol, **VOL_TOL)
        for sim in (True, False):
            vol, disp, mean = S.build_gwc_volume_with_disparity(dev(L), dev(R), D, G, return_mean=True, match_similarity=sim)
            close(vol, ovol, **VOL_TOL)
            close(mean, omean, **VOL_TOL)
            close(disp, oracle.soft_argmin(omean, sim), **DISP_TOL)


def test_volume_valu_kernels_all_epilogues(S, oracle):
    """The pipelined VALU volume kernel -- (16 disparities, 12 waves) for D > 64, (8, 8) below -- against the oracle: store and
    fused arg-min epilogues."""
    rng = np.random.default_rng(3)
    with S._native.test_options(corr_mfma=0):
        for (B, C, H, W, D, G) in ((1, 32, 5, 960, 192, 8), (2, 12, 6, 520, 150, 4), (1, 6, 9, 260, 97, 2), (1, 8, 4, 300, 48, 4)):
            L = rng.standard_normal((B, C, H, W), dtype=np.float32)
            R = rng.standard_normal((B, C, H, W), dtype=np.float32)
            close(S.build_gwc_volume(dev(L), dev(R), D, G), oracle.gwc_volume(L, R, D, G), **VOL_TOL)
            ref = oracle.corr_volume(L, R, D)
            close(S.CostVolume(D)(dev(L), dev(R)), ref, **VOL_TOL)
            close(S.corr_soft_argmin(dev(L), dev(R), D), oracle.soft_argmin(ref), **DISP_TOL)
            v, d = S.corr_volume_with_disparity(dev(L), dev(R), D)
            close(v, ref, **VOL_TOL)
            close(d, oracle.soft_argmin(ref), **DISP_TOL)


def test_volume_errors_and_edges(S):
    L = torch.randn(1, 6, 3, 8, device="cuda")
    with pytest.raises(NotImplementedError):
        S.CostVolume(4, "concated")(L, L)            # build_cost_volume.py:49-50
    with pytest.raises(AssertionError):
        S.build_gwc_volume(L, L, 4, 4)               # GWC_Cost_Volume.py:7
    with pytest.raises(RuntimeError):
        S.CostVolume(4)(L.cpu(), L.cpu())            # no CPU fallback
    with pytest.raises(TypeError):
        S.CostVolume(4)(L.double(), L.double())
    e = torch.empty(0, 6, 3, 8, device="cuda")
    assert S.CostVolume(4)(e, e).shape == (0, 4, 3, 8)
    assert S.build_gwc_volume(e, e, 4, 2).shape == (0, 2, 4, 3, 8)
    # non-contiguous inputs are accepted like in the reference
    Lt = torch.randn(1, 3, 8, 6, device="cuda").permute(0, 3, 1, 2)
    v = S.CostVolume(2)(Lt, Lt)
    v2 = S.CostVolume(2)(Lt.contiguous(), Lt.contiguous())
    assert torch.equal(v, v2)


# ------------------------------------------------------------------ soft-argmin (A4)
def test_soft_argmin_golden(S, golden):
    g = golden("g2_soft_argmin")
    vol = dev(g["vol"])
    close(S.DisparityEstimation(12, True)(vol), g["disp_sim"], **DISP_TOL)
    close(S.DisparityEstimation(12, False)(vol), g["disp_cost"], **DISP_TOL)
    close(S.DisparityEstimation(48, True)(vol), g["disp_other_maxdisp"], **DISP_TOL)
    d, p = S.DisparityEstimationWithProb(12, True)(vol)
    close(d, g["disp_withprob"], **DISP_TOL)
    close(p, g["prob"], rtol=0, atol=2e-6)
    close(S.DisparityEstimation(24, True)(dev(g["vol_big"])), g["disp_big"], **DISP_TOL)
    close(S.disparity_regression(dev(g["prob"]), 12), g["disp_withprob"], **DISP_TOL)
    with pytest.raises(AssertionError):
        S.DisparityEstimation(12)(vol[0])            # estimation.py:15


@pytest.mark.parametrize("shape", [(2, 64, 9, 40), (1, 192, 5, 37), (3, 7, 1, 5), (1, 1, 4, 4), (2, 192, 6, 260), (1, 48, 3, 100),
                                   (1, 97, 2, 1030), (1, 200, 2, 64)])
def test_soft_argmin_vs_oracle(S, oracle, shape):
    rng = np.random.default_rng(7)
    vol = (4.0 * rng.standard_normal(shape)).astype(np.float32)
    for sim in (True, False):
        close(S.DisparityEstimation(shape[1], sim)(dev(vol)), oracle.soft_argmin(vol, sim), **DISP_TOL)
    d, p = S.DisparityEstimationWithProb(shape[1])(dev(vol))
    od, op = oracle.soft_argmin(vol, True, with_prob=True)
    close(d, od, **DISP_TOL)
    close(p, op, rtol=0, atol=2e-6)
    d2, p2 = S.DisparityEstimationWithProb(shape[1], False)(dev(vol))
    od2, op2 = oracle.soft_argmin(vol, False, with_prob=True)
    close(d2, od2, **DISP_TOL)
    close(p2, op2, rtol=0, atol=2e-6)


@pytest.mark.parametrize("shape", [(2, 8, 24, 5, 40), (1, 3, 10, 3, 37), (1, 4, 192, 2, 64)])
def test_group_mean_soft_argmin(S, oracle, shape):
    rng = np.random.default_rng(11)
    vg = (3.0 * rng.standard_normal(shape)).astype(np.float32)
    mean = oracle.group_mean(vg)
    d, m = S.group_mean_soft_argmin(dev(vg), return_mean=True)
    close(m, mean, **VOL_TOL)
    close(d, oracle.soft_argmin(mean), **DISP_TOL)
    close(S.group_mean_soft_argmin(dev(vg), match_similarity=False), oracle.soft_argmin(mean, False), **DISP_TOL)


@pytest.mark.parametrize("shape", [(2, 32, 5, 300, 64, 8), (1, 32, 3, 960, 192, 8), (1, 6, 3, 37, 9, 3), (1, 8, 2, 40, 200, 2),
                                   (2, 24, 4, 516, 100, 4)])
def test_gwc_volume_with_disparity(S, oracle, shape):
    """A3 + group mean + A4 in one call (volume written once, never read back) == the separate ops == oracle,
    on the pipelined fast path (W % 4 == 0, D <= 192) and on the generic fallbacks."""
    B, C, H, W, D, G = shape
    rng = np.random.default_rng(29)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32) * 2
    R = rng.standard_normal((B, C, H, W), dtype=np.float32) * 2
    ovol = oracle.gwc_volume(L, R, D, G)
    omean = oracle.group_mean(ovol)
    for sim in (True, False):
        vol, disp, mean = S.build_gwc_volume_with_disparity(dev(L), dev(R), D, G, return_mean=True, match_similarity=sim)
        close(vol, ovol, **VOL_TOL)
        close(mean, omean, **VOL_TOL)
        close(disp, oracle.soft_argmin(omean, sim), **DISP_TOL)
        d2 = S.group_mean_soft_argmin(S.build_gwc_volume(dev(L), dev(R), D, G), match_similarity=sim)
        close(disp, host(d2), **DISP_TOL)
    vol, disp = S.build_gwc_volume_with_disparity(dev(L), dev(R), D, G)
    close(vol, ovol, **VOL_TOL)
    close(disp, oracle.soft_argmin(omean), **DISP_TOL)
    cv, cd = S.corr_volume_with_disparity(dev(L), dev(R), D)
    close(cv, oracle.corr_volume(L, R, D), **VOL_TOL)
    close(cd, oracle.soft_argmin(oracle.corr_volume(L, R, D)), **DISP_TOL)


@pytest.mark.parametrize("mfma", ["1", "0"])
@pytest.mark.parametrize("shape", [(2, 32, 5, 300, 64), (1, 32, 2, 960, 192), (1, 6, 3, 37, 9), (1, 8, 2, 40, 200)])
def test_fused_corr_soft_argmin(S, oracle, shape, mfma):
    with S._native.test_options(corr_mfma=int(mfma)):   # 1: the attention v5 kernel (identity I3); 0: VALU arg-min epilogue
        _fused_corr_soft_argmin(S, oracle, shape)


def _fused_corr_soft_argmin(S, oracle, shape):
    B, C, H, W, D = shape
    rng = np.random.default_rng(13)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32) * 2
    R = rng.standard_normal((B, C, H, W), dtype=np.float32) * 2
    ref = oracle.soft_argmin(oracle.corr_volume(L, R, D))
    close(S.corr_soft_argmin(dev(L), dev(R), D), ref, **DISP_TOL)
    two_step = S.DisparityEstimation(D)(S.CostVolume(D)(dev(L), dev(R)))
    close(S.corr_soft_argmin(dev(L), dev(R), D), host(two_step), **DISP_TOL)
    refc = oracle.soft_argmin(oracle.corr_volume(L, R, D), False)
    close(S.corr_soft_argmin(dev(L), dev(R), D, match_similarity=False), refc, **DISP_TOL)


# ------------------------------------------------------------------ local lookups (A5/A6/A11 sampler)
def test_local_lookup_golden(S, golden):
    g = golden("g3_local_lookup")
    vol, disp = dev(g["vol"]), dev(g["disp"])
    for (r, s) in ((2, 20), (4, 10), (2.1, 20)):
        for cv in (False, True):
            close(S.LocalCostVolume(r, s)(vol, disp, cv), g[f"fixed_r{r}_S{s}_cv{int(cv)}"], **DISP_TOL)
    for (gm, s) in ((1, 10), (0.5, 20)):
        for cv in (False, True):
            close(S.Variance_Based_Local_Cost_Volume(gm, s)(vol, disp, cv), g[f"var_g{gm}_S{s}_cv{int(cv)}"],
                  **DISP_TOL)
    close(S.build_cost_volume_from_volume(vol, dev(g["cand"])), g["sampled"], **VOL_TOL)


def test_local_lookup_vs_oracle(S, oracle):
    rng = np.random.default_rng(17)
    B, D, H, W = 2, 64, 11, 53
    vol = (2.0 * rng.standard_normal((B, D, H, W))).astype(np.float32)
    disp = (rng.random((B, 1, H, W)) * (D - 1)).astype(np.float32)
    for cv in (False, True):
        close(S.LocalCostVolume(2, 20)(dev(vol), dev(disp), cv), oracle.local_lookup(vol, disp, 2.0, 20, cv),
              **DISP_TOL)
        close(S.Variance_Based_Local_Cost_Volume(1, 10)(dev(vol), dev(disp), cv),
              oracle.variance_lookup(vol, disp, 1.0, 10, cv), **DISP_TOL)
        lo = rng.random((B, 1, H, W)).astype(np.float32) * 3
        up = rng.random((B, 1, H, W)).astype(np.float32) * 3
        bounds = np.concatenate([lo, up], 1)
        close(S.bounds_lookup(dev(vol), dev(disp), dev(bounds), 20, cv),
              oracle.bounds_lookup(vol, disp, lo, up, 20, cv), **DISP_TOL)
    close(S.uncertainty_volume(dev(vol), dev(disp)), oracle.uncertainty_volume(vol, disp), rtol=1e-4, atol=2e-4)


def test_identity_i5_invalid_pixels_return_input(S):
    D = 24
    vol = torch.randn(1, D, 8, 16, device="cuda")
    disp = torch.rand(1, 1, 8, 16, device="cuda") * (D - 1)
    out = S.LocalCostVolume(2, 20)(vol, disp, True)
    invalid = (disp - 2 < 0) | (disp + 2 >= D - 1)
    assert invalid.any()
    assert (out[invalid] - disp[invalid]).abs().max().item() < 1e-5


# ------------------------------------------------------------------ RAFT correlation (A8/A9/A10)
def test_raft_corr_golden(S, golden):
    g = golden("g4_raft_corr")
    f1, f2, coords = dev(g["f1"]), dev(g["f2"]), dev(g["coords"])
    B, C, H, W = g["f1"].shape
    close(S.CorrBlock1D.corr(f1, f2).reshape(B, H, W, W), g["allpairs"], **VOL_TOL)
    blk = S.CorrBlock1D(f1, f2, num_levels=4, radius=4)
    assert len(blk.corr_pyramid) == 5
    for i, lv in enumerate(blk.corr_pyramid):
        close(lv.reshape(B * H * W, -1), g[f"pyr{i}"], **VOL_TOL)
    close(blk(coords), g["lookup_r4_l4"], **VOL_TOL)
    close(S.CorrBlock1D(f1, f2, num_levels=2, radius=3)(coords), g["lookup_r3_l2"], **VOL_TOL)
    close(S.PytorchAlternateCorrBlock1D(f1, f2, 4, 4)(coords), g["alt_lookup_r4_l4"], rtol=1e-4, atol=1e-4)
    close(S.coords_grid(2, 3, 4, device="cuda"), g["coords_grid_3x4"], rtol=0, atol=0)
    s, m = S.bilinear_sampler(dev(g["bs_img"]), dev(g["bs_coords"]), mask=True)
    close(s, g["bs_out"], **VOL_TOL)
    close(m, g["bs_mask"], rtol=0, atol=0)
    with pytest.raises(NotImplementedError):
        S.AlternateCorrBlock(f1, f2)


@pytest.mark.parametrize("shape", [(1, 256, 3, 240), (2, 31, 2, 150), (1, 64, 1, 33), (1, 40, 2, 312), (1, 7, 1, 128), (1, 100, 2, 260)])
@pytest.mark.parametrize("f16p", [1, 0])
def test_allpairs_vs_oracle(S, oracle, shape, f16p):
    """A8 with the split-fp16 matrix phase (default) and the f32 MFMA (test option), incl. features of very different magnitude
    left and right (the low fp16 piece of a small value is subnormal: absolute accuracy)."""
    B, C, H, W = shape
    rng = np.random.default_rng(19)
    f1 = rng.standard_normal(shape, dtype=np.float32)
    f2 = rng.standard_normal(shape, dtype=np.float32)
    with S._native.test_options(corr_f16p=f16p):
        blk = S.CorrBlock1D(dev(f1), dev(f2), num_levels=4, radius=4)
        small = S.CorrBlock1D.corr(dev(f1 * 0.003), dev(f2 * 300)).reshape(B, H, W, W)
    ap = oracle.allpairs_corr(f1, f2)
    close(small, oracle.allpairs_corr((f1 * 0.003).astype(np.float32), (f2 * 300).astype(np.float32)), rtol=1e-5, atol=5e-5)
    pyr = oracle.corr_pyramid(ap, 4)
    for i in range(5):
        close(blk.corr_pyramid[i].reshape(pyr[i].shape), pyr[i], rtol=1e-5, atol=5e-5)
    if W >> 3 >= 2:
        coords = np.stack([np.tile(np.arange(W, dtype=np.float32), (H, 1)) - rng.random((H, W)).astype(np.float32) * 20,
                           np.zeros((H, W), np.float32)], 0)[None].repeat(B, 0)
        close(blk(dev(coords)), oracle.corr_lookup(pyr, coords, 4, 4), rtol=1e-5, atol=5e-5)


# ------------------------------------------------------------------ warp / upsample (A12/A13)
def test_disp_warp_golden(S, golden):
    g = golden("g6_disp_warp")
    for pm in ("border", "zeros"):
        w, m = S.disp_warp(dev(g["img"]), dev(g["disp"]), pm)
        close(w, g[f"warped_{pm}"], **VOL_TOL)
        # the valid mask thresholds grid_sample(ones) at 0.9999 (disparity_warper.py:101-103): a pixel may differ from the
        # reference only where that sampled value sits within 1e-5 of the threshold (1-ulp noise of the coordinate math)
        diff = host(m) != g[f"mask_{pm}"]
        near = np.abs(_sampled_ones(g["disp"], *g["img"].shape[2:]) - 0.9999) < 1e-5
        assert not (diff & ~np.broadcast_to(near[:, None], diff.shape)).any(), int((diff & ~near[:, None]).sum())
        assert diff.mean() < 0.01


def _sampled_ones(disp, H, W):
    """fp64 restatement of grid_sample(ones, zeros padding, align_corners=False) at (x - disp, y) normalised by (W-1),
    (H-1) -- the quantity disp_warp thresholds for its valid mask; [B,H,W]."""
    d = disp[:, 0].astype(np.float64)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    ux = (((2.0 * ((xs - d) / (W - 1)) - 1.0) + 1.0) * W - 1.0) / 2.0
    uy = (((2.0 * (ys / (H - 1)) - 1.0) + 1.0) * H - 1.0) / 2.0 + 0 * d
    fx, fy = np.floor(ux), np.floor(uy)
    ax, ay = ux - fx, uy - fy
    tot = np.zeros_like(d)
    for jy in (0, 1):
        for jx in (0, 1):
            ok = (fy + jy >= 0) & (fy + jy <= H - 1) & (fx + jx >= 0) & (fx + jx <= W - 1)
            tot += ok * (ax if jx else 1.0 - ax) * (ay if jy else 1.0 - ay)
    return tot


def test_upsample_golden(S, golden, oracle):
    g = golden("g7_upsample")
    close(S.upsample_convex8(dev(g["disp"]), dev(g["mask8"])), g["up8"], **VOL_TOL)
    close(S.upsample_flow(dev(g["flow"]), dev(g["mask4"]), 2), g["upflow4"], **VOL_TOL)
    close(S.upsample_simple8(dev(g["disp"])), g["simple8"], **VOL_TOL)
    close(S.upflow8(dev(np.concatenate([g["disp"], -g["disp"]], 1))), g["upflow8"], **VOL_TOL)
    # wider than one 64-pixel tile, two channels, factor 8 and a generic factor
    rng = np.random.default_rng(23)
    d = rng.random((1, 2, 3, 150)).astype(np.float32) * 30
    for f in (8, 4, 3):
        mk = rng.standard_normal((1, 9 * f * f, 3, 150)).astype(np.float32)
        close(S.upsample_convex(dev(d), dev(mk), f), oracle.upsample_convex(d, mk, f), **VOL_TOL)


def test_resize_vs_oracle(S, oracle, golden):
    g = golden("g5_gru_updater")
    h, w = g["disp0"].shape[2:]
    close(S.resize_bilinear(dev(g["left"]), (h, w), False), g["left_small"], **VOL_TOL)
    x = np.random.default_rng(5).standard_normal((2, 3, 9, 14)).astype(np.float32)
    close(S.resize_bilinear(dev(x), (18, 28), True), oracle.resize_bilinear(x, (18, 28), True), **VOL_TOL)
    close(S.resize_bilinear(dev(x), (5, 9), False), oracle.resize_bilinear(x, (5, 9), False), **VOL_TOL)


# ------------------------------------------------------------------ cross-attention (A7, build-defined)
@pytest.mark.parametrize("cfg", [(2, 16, 5, 40, 12), (1, 32, 3, 200, 192), (1, 48, 2, 70, 33), (1, 40, 2, 300, 230),
                                 (1, 256, 2, 312, 64), (1, 8, 1, 64, 1), (1, 32, 1, 130, 320), (1, 20, 1, 96, 32),
                                 (2, 32, 3, 960, 192), (1, 24, 2, 516, 250)])
@pytest.mark.parametrize("variant", ["auto", "v4", "v2", "v5", "v7"])
def test_cross_attention(S, oracle, cfg, variant):
    # v2: per-tile MFMA kernel with an LDS score sheet (unaligned / ragged shapes, 256 < D <= 320)
    # auto: v9 (split-fp16 matrix cores) for C <= 32 with value = right features, else v4; v5: the fp32 online-softmax kernel;
    # v7: the split-bf16 form v9 replaced as the default
    with S._native.test_options(att_kernel={"auto": 0, "v4": 4, "v2": 2, "v5": 5, "v7": 7}[variant]):
        _cross_attention(S, oracle, cfg)


def test_cross_attention_band_limit(S):
    """D beyond the LDS score sheet of the widest kernel (D > 320) is refused loudly, not served by a slow path."""
    x = torch.randn(1, 8, 1, 64, device="cuda")
    with pytest.raises(S._native.StereoHipError, match="supported range"):
        S.CrossAttentionCostVolume(352)(x, x)


def _cross_attention_fp64(L, R, V, D, mask, scale):
    """The function oracle_cross_attention restates (oracle/stereo_oracle.c: A7), in fp64 (torch CPU): the yardstick for the
    rounding noise of BOTH fp32 implementations at long bands."""
    L, R, V = (torch.from_numpy(np.asarray(t)).double() for t in (L, R, V))
    B, C, H, W = L.shape
    sc = 1.0 / np.sqrt(C) if scale == "sqrt_c" else 1.0 / C
    logits = torch.zeros(B, D, H, W, dtype=torch.float64)
    vals = torch.zeros(B, V.shape[1], D, H, W, dtype=torch.float64)
    for d in range(min(D, W)):
        logits[:, d, :, d:] = (L[:, :, :, d:] * R[:, :, :, :W - d]).sum(1) * sc
        vals[:, :, d, :, d:] = V[:, :, :, :W - d]
    if mask == "neginf":
        w = torch.arange(W).view(1, 1, 1, W)
        logits = logits.masked_fill(torch.arange(D).view(1, D, 1, 1) > w, float("-inf"))
    p = torch.softmax(logits, 1)
    disp = (p * torch.arange(D, dtype=torch.float64).view(1, D, 1, 1)).sum(1)
    return (p.unsqueeze(1) * vals).sum(2).numpy(), disp.numpy()


@pytest.mark.parametrize("cfg", [(1, 32, 1, 300, 256), (1, 32, 1, 330, 320), (1, 24, 2, 516, 250), (1, 40, 1, 330, 300)])
def test_cross_attention_long_bands_against_fp64(S, oracle, cfg):
    """D > 192: the mean |disparity error| of two fp32 implementations against each other grows with the band (summation-order
    noise of a D-term expectation), so the bar is anchored in fp64 instead of scaled with D: the HIP kernels may be at most
    1.5 x as far from the fp64 statement as the C oracle is (plus 2e-5 px), and inside the north-star 1e-3 bar."""
    B, C, H, W, D = cfg
    rng = np.random.default_rng(30)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32)
    R = rng.standard_normal((B, C, H, W), dtype=np.float32)
    for mask in ("zero", "neginf"):
        for scale in ("sqrt_c", "mean"):
            o64, d64 = _cross_attention_fp64(L, R, R, D, mask, scale)
            oo, od = oracle.cross_attention(L, R, None, D, mask, scale)
            o, d = S.CrossAttentionCostVolume(D, mask, scale)(dev(L), dev(R))
            e_hip, e_orc = np.abs(host(d) - d64), np.abs(od - d64)
            assert e_hip.max() <= 1e-3 and e_hip.mean() <= 1.5 * e_orc.mean() + 2e-5, (e_hip.max(), e_hip.mean(), e_orc.mean())
            eo_hip, eo_orc = np.abs(host(o) - o64).max(), np.abs(oo - o64).max()
            assert eo_hip <= 1.5 * eo_orc + 2e-5, (eo_hip, eo_orc)


def _cross_attention(S, oracle, cfg):
    B, C, H, W, D = cfg
    rng = np.random.default_rng(29)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32)
    R = rng.standard_normal((B, C, H, W), dtype=np.float32)
    V = rng.standard_normal((B, 20 if C != 40 else 70, H, W), dtype=np.float32)
    for mask in ("zero", "neginf"):
        for scale in ("sqrt_c", "mean"):
            o, d = S.CrossAttentionCostVolume(D, mask, scale)(dev(L), dev(R), dev(V))
            oo, od = oracle.cross_attention(L, R, V, D, mask, scale)
            close(o, oo, rtol=1e-4, atol=5e-5)
            # (mean bar: D <= 192 only -- longer bands are anchored in fp64, test_cross_attention_long_bands_against_fp64)
            close(d, od, rtol=0, atol=1e-3, mean_atol=2.5e-4 if D <= 192 else None)
    # value defaults to the right features (one staged tile serves as K and V when C <= 32)
    o, d = S.CrossAttentionCostVolume(D, "neginf", "sqrt_c")(dev(L), dev(R))
    oo, od = oracle.cross_attention(L, R, None, D, "neginf", "sqrt_c")
    close(o, oo, rtol=1e-4, atol=5e-5)
    close(d, od, rtol=0, atol=1e-3, mean_atol=2.5e-4 if D <= 192 else None)
    # I3: mask='zero', scale='mean' reproduces CostVolume + DisparityEstimation of the HIP path itself
    _, d = S.CrossAttentionCostVolume(D, "zero", "mean")(dev(L), dev(R), need_out=False)
    close(d, host(S.DisparityEstimation(D)(S.CostVolume(D)(dev(L), dev(R)))), rtol=0, atol=1e-3,
          mean_atol=2.5e-4 if D <= 192 else None)


@pytest.mark.parametrize("kern", [0, 5])
@pytest.mark.parametrize("nw,tpw", [(4, 0), (8, 0), (8, 1), (4, 3), (6, 0), (6, 2), (8, 3), (8, 5)])
def test_cross_attention_online_workgroup_shapes(S, oracle, nw, tpw, kern):
    """Online-softmax kernels (0: split-bf16, 5: fp32) with 4 / 6 / 8 waves (128 / 192 / 256 queries) per workgroup tile (the
    fp32 one has no 6-wave shape and takes its own pick there) and 1 / 2 / 3 / auto tiles per persistent workgroup, incl.
    partial last tiles, every band size class (NT = 3, 5, 7, 9), both masks, disparity-only mode.  (8, 3) / (8, 5): runs of
    256-query tiles that continue along a row, start a new row and start a workgroup mid-row -- the key RING of the default
    kernel (a continuing tile stages only its 256 new keys) in all its transitions, incl. a ragged last tile."""
    rng = np.random.default_rng(31)
    for (B, C, H, W, D) in ((1, 32, 2, 960, 192), (2, 16, 3, 200, 40), (1, 8, 2, 516, 100), (1, 32, 1, 300, 256), (1, 5, 2, 36, 7),
                            (1, 24, 3, 772, 170)):
        L = rng.standard_normal((B, C, H, W), dtype=np.float32)
        R = rng.standard_normal((B, C, H, W), dtype=np.float32)
        for mask in ("zero", "neginf"):
            oo, od = oracle.cross_attention(L, R, None, D, mask, "sqrt_c")
            with S._native.test_options(att_waves=nw, tiles_per_wg=tpw, att_kernel=kern):
                o, d = S.CrossAttentionCostVolume(D, mask, "sqrt_c")(dev(L), dev(R))
                _, d2 = S.CrossAttentionCostVolume(D, mask, "sqrt_c")(dev(L), dev(R), need_out=False)
            close(o, oo, rtol=1e-4, atol=5e-5)
            close(d, od, rtol=0, atol=1e-3, mean_atol=2.5e-4 * max(1.0, D / 192.0))
            close(d2, od, rtol=0, atol=1e-3, mean_atol=2.5e-4 * max(1.0, D / 192.0))


def test_cross_attention_key_ring_random_runs(S, oracle):
    """The default C <= 32 kernel's key ring (256-query tiles, D > 128: a tile that continues its predecessor's row stages only
    its 256 new keys) under random widths, bands, channel counts and lengths of the persistent workgroups' tile runs."""
    rng = np.random.default_rng(77)
    for it in range(12):
        B, H = int(rng.integers(1, 3)), int(rng.integers(1, 4))
        C = int(rng.integers(1, 33))
        W = 4 * int(rng.integers(130, 340))
        D = int(rng.integers(161, 193))
        tpw = int(rng.choice([0, 1, 2, 3, 4, 7]))   # 1: every tile restarts the ring (mid-row halo loads); 3, 7: rows end mid-run
        mask = str(rng.choice(["zero", "neginf"]))
        L = rng.standard_normal((B, C, H, W), dtype=np.float32)
        R = rng.standard_normal((B, C, H, W), dtype=np.float32)
        if it >= 8:   # very different magnitudes left and right: the low fp16 piece of a small value is subnormal (absolute accuracy)
            L, R = (L * 0.004).astype(np.float32), (R * 250).astype(np.float32)
        oo, od = oracle.cross_attention(L, R, None, D, mask, "sqrt_c")
        with S._native.test_options(att_waves=8, tiles_per_wg=tpw):
            o, d = S.CrossAttentionCostVolume(D, mask, "sqrt_c")(dev(L), dev(R))
            _, d2 = S.CrossAttentionCostVolume(D, mask, "sqrt_c")(dev(L), dev(R), need_out=False)
        close(o, oo, rtol=1e-4, atol=5e-5 * max(1.0, float(np.abs(R).max()) / 4))   # (absolute part: relative to the values summed)
        close(d, od, rtol=0, atol=1e-3, mean_atol=2.5e-4)
        close(d2, od, rtol=0, atol=1e-3, mean_atol=2.5e-4)


# ------------------------------------------------------------------ GRU gate math
def test_gru_pointwise(S, oracle):
    from stereoformer_amd import _native as N
    rng = np.random.default_rng(31)
    for n in (1000, 1023):
        a = [rng.standard_normal(n).astype(np.float32) * 2 for _ in range(5)]
        h, zc, qc, cz, cq = [dev(x) for x in a]
        out = torch.empty_like(h)
        N.call("stereo_gru_blend_f32", N.ptr(h), N.ptr(zc), N.ptr(qc), N.ptr(cz), N.ptr(cq), N.ptr(out), n, 0,
               N.stream_ptr())
        close(out, oracle.gru_blend(*a), rtol=1e-5, atol=2e-6)
        N.call("stereo_gru_blend_f32", N.ptr(h), N.ptr(zc), N.ptr(qc), N.ptr(None), N.ptr(None), N.ptr(out), n, 1,
               N.stream_ptr())
        close(out, oracle.gru_blend(a[0], a[1], a[2], softsign=True), rtol=1e-5, atol=2e-6)
        N.call("stereo_gru_reset_f32", N.ptr(h), N.ptr(zc), N.ptr(cz), N.ptr(out), n, N.stream_ptr())
        close(out, oracle.gru_reset(a[0], a[1], a[3]), rtol=1e-5, atol=2e-6)


def test_gru_pointwise_fp16(S):
    """fp16-in / fp16-out gate kernels (update block under mixed_precision): against the fp32 formula on the same
    (already fp16-rounded) inputs, to about one fp16 ulp; NCHW, channels-last and odd sizes (scalar path)."""
    torch.manual_seed(3)
    for shape, cl in (((2, 16, 9, 12), False), ((2, 16, 9, 12), True), ((1, 3, 5, 7), False)):
        mk = lambda: (torch.randn(shape, device="cuda") * 2).half()
        h, a, b, c1, c2 = mk(), mk(), mk(), mk(), mk()
        if cl:
            h, a, b, c1, c2 = [t.contiguous(memory_format=torch.channels_last) for t in (h, a, b, c1, c2)]
        r = S.gru_reset(h, a, c1)
        assert r.dtype == torch.float16 and r.stride() == h.stride()
        ref = torch.sigmoid(a.float() + c1.float()) * h.float()
        assert (r.float() - ref).abs().max().item() < 4e-3
        for soft in (False, True):
            o = S.gru_blend(h, a, b, c1, c2, softsign=soft)
            z = torch.sigmoid(a.float() + c1.float())
            qx = b.float() + c2.float()
            q = qx / (1 + qx.abs()) if soft else torch.tanh(qx)
            ref = (1 - z) * h.float() + z * q
            assert o.dtype == torch.float16 and (o.float() - ref).abs().max().item() < 6e-3
        z = torch.sigmoid(a.float())
        assert (S.gru_blend(h, a, b).float() - ((1 - z) * h.float() + z * torch.tanh(b.float()))).abs().max().item() < 6e-3


# ------------------------------------------------------------------ BASELINE configs
def test_cfg1_chain_vs_oracle(S, oracle):
    """configs[0]: 1 pair 384x768 D=64: corr -> soft-argmin -> LocalCostVolume(r=2,S=20,consider_valid)."""
    g = torch.Generator().manual_seed(0)
    L = torch.randn(1, 32, 384, 768, generator=g)
    R = torch.randn(1, 32, 384, 768, generator=g)
    vol = S.CostVolume(64)(L.cuda(), R.cuda())
    disp = S.DisparityEstimation(64)(vol)
    d2 = S.LocalCostVolume(2, 20)(vol, disp.unsqueeze(1), True)
    ovol = oracle.corr_volume(L.numpy(), R.numpy(), 64)
    odisp = oracle.soft_argmin(ovol)
    od2 = oracle.local_lookup(ovol, odisp[:, None], 2.0, 20, True)
    close(vol, ovol, **VOL_TOL)
    close(disp, odisp, rtol=0, atol=1e-3)   # the north-star bar; measured ~1e-4
    assert np.abs(host(disp) - odisp).mean() < 1e-4
    close(d2, od2, rtol=0, atol=1e-3)


def test_cfg2_full_size_identities(S, oracle):
    """configs[1] at full size (B=8, 960x540, D=192, G=8): the 25.5 GB group-wise volume is
    checked through identities (I1 group-mean == correlation volume, fused == two-step) and
    against the oracle on three full rows."""
    B, C, H, W, D, G = 8, 32, 540, 960, 192, 8
    g = torch.Generator(device="cuda").manual_seed(1)
    L = torch.randn(B, C, H, W, device="cuda", generator=g)
    R = torch.randn(B, C, H, W, device="cuda", generator=g)
    volg = S.build_gwc_volume(L, R, D, G)
    assert volg.shape == (B, G, D, H, W)
    disp_g, mean = S.group_mean_soft_argmin(volg, return_mean=True)
    rows = [0, 269, 539]
    for b in (0, 7):
        Ls, Rs = host(L[b:b + 1, :, rows]), host(R[b:b + 1, :, rows])
        close(volg[b:b + 1, :, :, rows], oracle.gwc_volume(Ls, Rs, D, G), **VOL_TOL)
        close(disp_g[b:b + 1, rows], oracle.soft_argmin(oracle.corr_volume(Ls, Rs, D)), rtol=0, atol=1e-3)
    del volg
    corr = S.CostVolume(D)(L, R)
    assert (corr - mean).abs().max().item() < 2e-5          # I1
    del mean
    # zero fill where w < d
    assert corr[:, 100, :, :100].abs().max().item() == 0.0
    disp = S.DisparityEstimation(D)(corr)
    assert (disp - disp_g).abs().max().item() < 1e-3
    volf, dispf, meanf = S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True)
    assert (meanf - corr).abs().max().item() < 2e-5 and (dispf - disp_g).abs().max().item() < 1e-3
    assert torch.equal(volf[3, :, 150:160], S.build_gwc_volume(L[3:4], R[3:4], D, G)[0, :, 150:160])
    del volf, meanf
    fused = S.corr_soft_argmin(L, R, D)
    assert (fused - disp).abs().max().item() < 1e-3
    assert (fused - disp).abs().mean().item() < 1e-4
    d2 = S.LocalCostVolume(2, 20)(corr, disp.unsqueeze(1), True)
    assert torch.isfinite(d2).all()
    close(d2[0:1, :, rows], oracle.local_lookup(host(corr[0:1, :, rows]), host(disp[0:1, None, rows]), 2.0, 20, True),
          **DISP_TOL)


# ------------------------------------------------------------------ edge cases across the whole surface
def test_empty_and_degenerate_inputs(S):
    z = lambda *s: torch.zeros(*s, device="cuda")
    # zero-sized batches launch nothing and return correctly shaped outputs
    assert S.DisparityEstimation(4)(z(0, 4, 3, 8)).shape == (0, 3, 8)
    d, p = S.DisparityEstimationWithProb(4)(z(0, 4, 3, 8))
    assert d.shape == (0, 3, 8) and p.shape == (0, 4, 3, 8)
    assert S.LocalCostVolume(2, 5)(z(0, 4, 3, 8), z(0, 1, 3, 8)).shape == (0, 1, 3, 8)
    assert S.Variance_Based_Local_Cost_Volume(1, 5)(z(0, 4, 3, 8), z(0, 1, 3, 8), True).shape == (0, 1, 3, 8)
    assert S.upsample_convex8(z(0, 1, 3, 4), z(0, 576, 3, 4)).shape == (0, 1, 24, 32)
    assert S.disp_warp(z(0, 3, 4, 5), z(0, 1, 4, 5))[0].shape == (0, 3, 4, 5)
    o, dd = S.CrossAttentionCostVolume(4)(z(0, 8, 3, 8), z(0, 8, 3, 8))
    assert o.shape == (0, 8, 3, 8) and dd.shape == (0, 3, 8)
    assert S.group_mean_soft_argmin(z(0, 2, 4, 3, 8)).shape == (0, 3, 8)
    v0, d0 = S.build_gwc_volume_with_disparity(z(0, 8, 3, 8), z(0, 8, 3, 8), 4, 2)
    assert v0.shape == (0, 2, 4, 3, 8) and d0.shape == (0, 3, 8)
    assert S.gru_blend(z(0, 4, 2, 2), z(0, 4, 2, 2), z(0, 4, 2, 2)).shape == (0, 4, 2, 2)
    # a single disparity plane: softmax over one element -> disparity 0 everywhere
    v1 = torch.randn(2, 1, 3, 8, device="cuda")
    assert S.DisparityEstimation(1)(v1).abs().max().item() == 0.0
    # H == 1, W == 1 volumes
    L = torch.randn(1, 4, 1, 1, device="cuda")
    assert torch.allclose(S.CostVolume(1)(L, L), (L * L).mean(1, keepdim=True))
    # an all-equal volume gives the mean candidate
    v = torch.zeros(1, 9, 2, 4, device="cuda")
    assert torch.allclose(S.DisparityEstimation(9)(v), torch.full((1, 2, 4), 4.0, device="cuda"), atol=1e-5)
    # shape errors are raised on the host, before any launch
    with pytest.raises(ValueError):
        S.LocalCostVolume(2, 5)(v, torch.zeros(1, 1, 2, 5, device="cuda"))
    with pytest.raises(RuntimeError):
        S.upsample_convex8(torch.zeros(1, 1, 2, 2, device="cuda"), torch.zeros(1, 100, 2, 2, device="cuda"))
    with pytest.raises(RuntimeError, match="forward-only"):   # the fused inference entry points refuse grad loudly
        S.corr_soft_argmin(torch.randn(1, 2, 2, 4, device="cuda", requires_grad=True), torch.randn(1, 2, 2, 4, device="cuda"), 2)


def test_streams_and_determinism(S):
    """Launches go to the caller's current stream and results are bit-reproducible."""
    L = torch.randn(2, 32, 9, 300, device="cuda")
    R = torch.randn(2, 32, 9, 300, device="cuda")
    ref = S.build_gwc_volume(L, R, 64, 8)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        out = S.build_gwc_volume(L, R, 64, 8)
        d1 = S.group_mean_soft_argmin(out)
    st.synchronize()
    assert torch.equal(out, ref)
    assert torch.equal(d1, S.group_mean_soft_argmin(ref))
    assert torch.equal(S.corr_soft_argmin(L, R, 64), S.corr_soft_argmin(L, R, 64))


def test_cfg3_full_size_identity(S, oracle):
    """configs[2] at full size (B=8, 960x540, D=192, C=32): the MFMA cross-attention against identity I3
    (mask='zero', scale='mean' == correlation volume + soft-argmin) and against the oracle on two rows."""
    B, C, H, W, D = 8, 32, 540, 960, 192
    g = torch.Generator(device="cuda").manual_seed(3)
    L = torch.randn(B, C, H, W, device="cuda", generator=g)
    R = torch.randn(B, C, H, W, device="cuda", generator=g)
    out, disp = S.CrossAttentionCostVolume(D, "zero", "mean")(L, R)
    fused = S.corr_soft_argmin(L, R, D)
    assert (disp - fused).abs().max().item() < 1e-3 and (disp - fused).abs().mean().item() < 1e-4
    rows = [0, 377]
    for b in (0, 5):
        Ls, Rs = host(L[b:b + 1, :, rows]), host(R[b:b + 1, :, rows])
        oo, od = oracle.cross_attention(Ls, Rs, None, D, "zero", "mean")
        close(disp[b:b + 1, rows], od, **DISP_TOL)
        close(out[b:b + 1, :, rows], oo, rtol=1e-4, atol=5e-5)
    _, dn = S.CrossAttentionCostVolume(D, "neginf", "sqrt_c")(L, R, need_out=False)
    assert (dn <= torch.arange(W, device="cuda").view(1, 1, W) + 1e-3).all()   # mass only on in-image keys


def test_cfg5_kitti_shapes(S, oracle):
    """configs[4] geometry: KITTI 375x1242 padded to 384x1248 (divis_by=32), D=256 at full resolution and the
    1/4-scale attention [4,256,96,312] with a 64-disparity band."""
    pad = S.InputPadder((375, 1242), mode="kitti", divis_by=32)
    (img,) = pad.pad(torch.randn(1, 3, 375, 1242, device="cuda"))
    assert img.shape[-2:] == (384, 1248) and pad.unpad(img).shape[-2:] == (375, 1242)
    rng = np.random.default_rng(41)
    L = rng.standard_normal((1, 32, 3, 1248), dtype=np.float32)
    R = rng.standard_normal((1, 32, 3, 1248), dtype=np.float32)
    ref = oracle.corr_volume(L, R, 256)
    close(S.CostVolume(256)(dev(L), dev(R)), ref, **VOL_TOL)
    close(S.build_gwc_volume(dev(L), dev(R), 256, 8), oracle.gwc_volume(L, R, 256, 8), **VOL_TOL)
    close(S.corr_soft_argmin(dev(L), dev(R), 256), oracle.soft_argmin(ref), rtol=0, atol=1e-3, mean_atol=3.5e-4)
    f1 = rng.standard_normal((1, 256, 3, 312), dtype=np.float32)
    f2 = rng.standard_normal((1, 256, 3, 312), dtype=np.float32)
    o, d = S.CrossAttentionCostVolume(64, "neginf", "sqrt_c")(dev(f1), dev(f2))
    oo, od = oracle.cross_attention(f1, f2, None, 64, "neginf", "sqrt_c")
    close(o, oo, rtol=1e-4, atol=1e-4)
    close(d, od, **DISP_TOL)
    # flow_init from the attention disparity feeds the RAFT loop as (-disp, 0) (raft_stereo.py:163-164)
    blk = S.CorrBlock1D(dev(f1), dev(f2), 4, 4)
    coords = S.coords_grid(1, 3, 312, device="cuda")
    coords[:, :1] -= d.unsqueeze(1)
    assert blk(coords.contiguous()).shape == (1, 36, 3, 312)


# ------------------------------------------------------------------ registered ops / torch.compile
def test_module_surfaces_compile_without_graph_breaks(S, oracle):
    """The cfg-2 chain and the attention module traced by torch.compile(fullgraph=True) (aot_eager backend: Dynamo + AOT
    autograd tracing through the registered `stereo_hip::*` ops, no code generation): same numbers as the eager modules,
    and gradients flow through the ops' registered backward formulas."""
    import torch._dynamo
    rng = np.random.default_rng(91)
    L = dev(rng.standard_normal((1, 16, 4, 64), dtype=np.float32))
    R = dev(rng.standard_normal((1, 16, 4, 64), dtype=np.float32))
    est, look, att = S.DisparityEstimation(12), S.LocalCostVolume(2, 10), S.CrossAttentionCostVolume(12, "neginf", "sqrt_c")

    def chain(l, r):
        volg, disp, mean = S.build_gwc_volume_with_disparity(l, r, 12, 4, return_mean=True)
        d2 = est(S.CostVolume(12)(l, r))
        ref = look(mean, disp.unsqueeze(1), True)
        o, da = att(l, r)
        return volg, disp, d2, ref, o, da

    torch._dynamo.reset()
    compiled = torch.compile(chain, fullgraph=True, backend="aot_eager")
    with torch.no_grad():
        got, want = compiled(L, R), chain(L, R)
    for a, b in zip(got, want):
        assert torch.equal(a, b)

    def train_chain(l, r):
        vol = S.CostVolume(12)(l, r)
        d = est(vol).unsqueeze(1)
        return look(vol, d, False).sum() + S.build_gwc_volume(l, r, 12, 4).square().sum()

    la, ra = L.clone().requires_grad_(True), R.clone().requires_grad_(True)
    torch.autograd.backward(torch.compile(train_chain, fullgraph=True, backend="aot_eager")(la, ra))
    lb, rb = L.clone().requires_grad_(True), R.clone().requires_grad_(True)
    torch.autograd.backward(train_chain(lb, rb))
    close(la.grad, host(lb.grad), rtol=1e-5, atol=1e-5)
    close(ra.grad, host(rb.grad), rtol=1e-5, atol=1e-5)


def test_out_buffers_and_placement(S, oracle):
    """`out=` of the fused volume stage: caller-owned buffers are overwritten completely and returned; placed buffers
    (hbm.place_volume_stage_outputs) give the same numbers as fresh allocations; wrong buffers are refused on the host."""
    rng = np.random.default_rng(97)
    B, C, H, W, D, G = 1, 16, 3, 64, 20, 4
    L, R = dev(rng.standard_normal((B, C, H, W), dtype=np.float32)), dev(rng.standard_normal((B, C, H, W), dtype=np.float32))
    want = S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True)
    bufs, rep = S.hbm.place_volume_stage_outputs(L, R, D, G, max_tries=3, good_enough_tbs=1e9, arena=False)
    assert len(rep["probe_ms"]) == 3 and rep["chosen_ms"] == min(rep["probe_ms"])
    for b_ in bufs:
        b_.fill_(float("nan"))
    got = S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=bufs)
    for a, b_, w in zip(got, bufs, want):
        assert a.data_ptr() == b_.data_ptr() and torch.equal(a, w)
    # mapped-arena mode (the default): the volume is a window of a chunk-mapped arena whose other chunks are released afterwards
    free0 = torch.cuda.mem_get_info()[0]
    mbufs, mrep = S.hbm.place_volume_stage_outputs(L, R, D, G, good_enough_tbs=1e9, mapped_span_bytes=32 << 20, mapped_chunk_bytes=1 << 20)
    assert mrep["mode"] == "mapped arena" and mrep["searched_bytes"] == 33 << 20 and mrep["held_beyond_the_buffer_bytes"] < 1 << 20
    assert len(mrep["probe_ms_by_offset_GiB"]) >= 5 and mrep["chosen_offset_bytes"] % (2 << 20) == 0
    assert free0 - torch.cuda.mem_get_info()[0] < 24 << 20          # the other 32 chunks went back to the driver
    for b_ in mbufs:
        b_.fill_(float("nan"))
    got = S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=mbufs)
    for a, b_, w in zip(got, mbufs, want):
        assert a.data_ptr() == b_.data_ptr() and torch.equal(a, w)
    keep = mbufs[0].clone()
    del mbufs, got, a, b_                                             # the last window tensor unmaps the arena
    torch.cuda.synchronize()
    assert torch.equal(keep, want[0])
    # plain-arena mode: a window of one ordinary allocation (volume + slack) that stays allocated; same results
    abufs, arep = S.hbm.place_volume_stage_outputs(L, R, D, G, good_enough_tbs=1e9, arena_slack_bytes=16 << 20, arena_step_bytes=1 << 20,
                                                   mapped=False)
    assert arep["mode"] == "arena" and arep["held_beyond_the_buffer_bytes"] == 16 << 20
    assert arep["chosen_ms"] == min(arep["probe_ms_by_offset_GiB"].values()) or len(arep["probe_ms_by_offset_GiB"]) >= 5
    assert arep["chosen_offset_bytes"] % (1 << 20) == 0 and abufs[0].is_contiguous()
    for b_ in abufs:
        b_.fill_(float("nan"))
    got = S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=abufs)
    for a, b_, w in zip(got, abufs, want):
        assert a.data_ptr() == b_.data_ptr() and torch.equal(a, w)
    with pytest.raises(ValueError, match="out"):
        S.build_gwc_volume_with_disparity(L, R, D, G, out=(bufs[0], bufs[1], bufs[2][:, :-1]))
    # out= under grad: the same (placed) buffers serve every training step -- in-place results on history-free aliases
    cot = torch.randn(B, H, W, device="cuda")
    Lr, Rr = L.clone().requires_grad_(True), R.clone().requires_grad_(True)
    _, d0, m0 = S.build_gwc_volume_with_disparity(Lr, Rr, D, G, return_mean=True)
    g0 = torch.autograd.grad((d0 * cot).sum() + m0.square().sum(), (Lr, Rr))
    for step in range(2):
        v1, d1, m1 = S.build_gwc_volume_with_disparity(Lr, Rr, D, G, return_mean=True, out=bufs)
        assert v1.data_ptr() == bufs[0].data_ptr() and d1.data_ptr() == bufs[1].data_ptr() and m1.data_ptr() == bufs[2].data_ptr()
        assert d1.requires_grad and not bufs[1].requires_grad and bufs[1].grad_fn is None
        g1 = torch.autograd.grad((d1 * cot).sum() + m1.square().sum(), (Lr, Rr), retain_graph=step == 1)
        for a, b_ in zip(g1, g0):
            assert torch.equal(a, b_)
    S.build_gwc_volume_with_disparity(Lr, Rr, D, G, return_mean=True, out=bufs)      # overwrites the buffers of the retained graph
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        torch.autograd.grad((d1 * cot).sum() + m1.square().sum(), (Lr, Rr))
    with pytest.raises(RuntimeError, match="must not require grad"):
        S.build_gwc_volume_with_disparity(Lr, Rr, D, G, out=(bufs[0], bufs[1], bufs[2].clone().requires_grad_(True)))


@pytest.mark.parametrize("cfg", [(2, 32, 3, 520, 192), (1, 20, 2, 260, 70), (1, 7, 1, 64, 5), (1, 32, 2, 960, 130), (2, 16, 2, 100, 33)])
@pytest.mark.parametrize("f16p", [1, 0])
def test_corr_volume_matrix_phase_forms(S, oracle, cfg, f16p):
    """A1 (+ A4) on the matrix-core volume kernel with the split-fp16 matrix phase (default) and with the f32 MFMA (test hook):
    both against the oracle at the fp32 tolerances, fp32 and fp16 features."""
    B, C, H, W, D = cfg
    rng = np.random.default_rng(77)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32)
    R = rng.standard_normal((B, C, H, W), dtype=np.float32)
    with S._native.test_options(corr_f16p=f16p):
        close(S.CostVolume(D)(dev(L), dev(R)), oracle.corr_volume(L, R, D), **VOL_TOL)
        vol, disp = S.corr_volume_with_disparity(dev(L), dev(R), D)
        ov = oracle.corr_volume(L, R, D)
        close(vol, ov, **VOL_TOL)
        close(disp, oracle.soft_argmin(ov), rtol=0, atol=1e-3)
        Lh, Rh = L.astype(np.float16), R.astype(np.float16)
        close(S.CostVolume(D)(dev(Lh), dev(Rh)), oracle.corr_volume(Lh.astype(np.float32), Rh.astype(np.float32), D), **VOL_TOL)
