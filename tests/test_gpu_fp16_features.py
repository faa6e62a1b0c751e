"""GPU parity of the N2 feature hand-over: fp16 feature operands into the matrix-core kernels (A1 / A3 / A1+A4 /
A3+mean+A4 / A7 / A8) and the fused InstanceNorm2d + ReLU extractor epilogue.

The kernels widen fp16 features on the way into LDS and compute in fp32, so the ORACLE FED THE SAME fp16-ROUNDED INPUTS must
be reproduced at the fp32 tolerances (volumes 2e-5, disparities 1e-3 max / 2.5e-4 mean -- the north-star bar).  Against the
oracle on the UN-rounded fp32 inputs the difference is the fp16 rounding of the features, reported, not asserted tightly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VOL_TOL = dict(rtol=1e-5, atol=2e-5)


def close(a, b, rtol=0.0, atol=0.0, mean_atol=None):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)
    if mean_atol is not None and a.size:
        assert float(np.abs(a - b).mean()) <= mean_atol, float(np.abs(a - b).mean())


def disp_close(a, b, D=192):
    close(a, b, atol=1e-3, mean_atol=2.5e-4 * max(1.0, D / 192.0))


def h16(x):
    """fp32 numpy -> (fp16 CUDA tensor, the fp16-rounded values as fp32 numpy)."""
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda().half()
    return t, t.float().cpu().numpy()


@pytest.fixture(scope="module")
def S():
    import stereoformer_amd as s
    s._native.load()
    return s


@pytest.mark.parametrize("shape", [(1, 32, 4, 960, 192, 8), (2, 12, 5, 520, 150, 3), (1, 32, 3, 300, 64, 2), (1, 8, 4, 260, 97, 4),
                                   (2, 16, 2, 36, 7, 1), (1, 24, 2, 1300, 40, 12), (2, 32, 1, 8, 5, 16), (1, 16, 3, 1248, 256, 1)])
def test_volumes_fp16_features(S, oracle, shape):
    """A1, A3, A1+A4 and the fused cfg-2 volume stage (A3 + group mean + A4) on fp16 features == oracle on the rounded
    values; the fp16 entry point itself is exercised (a silent widening would pass too, so the C ABI is called directly)."""
    from stereoformer_amd import _native as N
    B, C, H, W, D, G = shape
    rng = np.random.default_rng(71)
    Lh, L = h16(rng.standard_normal((B, C, H, W), dtype=np.float32) * 2)
    Rh, R = h16(rng.standard_normal((B, C, H, W), dtype=np.float32) * 2)
    ocorr = oracle.corr_volume(L, R, D)
    vol = torch.empty(B, D, H, W, device="cuda")
    N.call("stereo_corr_volume_f16", N.ptr(Lh), N.ptr(Rh), N.ptr(vol), B, C, H, W, D, N.stream_ptr())   # no fp32 detour
    close(vol, ocorr, **VOL_TOL)
    close(S.CostVolume(D)(Lh, Rh), ocorr, **VOL_TOL)
    for sim in (True, False):
        v, d = S.corr_volume_with_disparity(Lh, Rh, D, match_similarity=sim)
        close(v, ocorr, **VOL_TOL)
        disp_close(d, oracle.soft_argmin(ocorr, sim), D)
        disp_close(S.corr_soft_argmin(Lh, Rh, D, sim), oracle.soft_argmin(ocorr, sim), D)
    if D <= 192 and C // G in (2, 4, 8, 16):
        ovol = oracle.gwc_volume(L, R, D, G)
        omean = oracle.group_mean(ovol)
        close(S.build_gwc_volume(Lh, Rh, D, G), ovol, **VOL_TOL)
        volg, disp, mean = S.build_gwc_volume_with_disparity(Lh, Rh, D, G, return_mean=True)
        close(volg, ovol, **VOL_TOL)
        close(mean, omean, **VOL_TOL)
        disp_close(disp, oracle.soft_argmin(omean), D)
        volg2, disp2 = S.build_gwc_volume_with_disparity(Lh, Rh, D, G)
        assert torch.equal(volg2, volg) and torch.equal(disp2, disp)
    # several tiles per persistent workgroup
    with N.test_options(tiles_per_wg=3):
        v, d = S.corr_volume_with_disparity(Lh, Rh, D)
    close(v, ocorr, **VOL_TOL)
    disp_close(d, oracle.soft_argmin(ocorr), D)


def test_fp16_outside_the_matrix_core_envelope_is_widened_on_the_host(S, oracle):
    """C > 32, ragged W, odd channels per group: the `_f16` entry point refuses (STEREO_ERR_UNSUPPORTED, no slow fp16 path) and
    the host layer widens the features -- same numbers as the oracle on the rounded values."""
    from stereoformer_amd import _native as N
    rng = np.random.default_rng(72)
    for (B, C, H, W, D, G) in ((1, 48, 2, 64, 24, 4), (1, 6, 3, 37, 9, 3), (1, 12, 2, 40, 10, 4)):
        Lh, L = h16(rng.standard_normal((B, C, H, W), dtype=np.float32))
        Rh, R = h16(rng.standard_normal((B, C, H, W), dtype=np.float32))
        vol = torch.empty(B, G, D, H, W, device="cuda")
        with pytest.raises(N.StereoHipError) as ei:
            N.call("stereo_gwc_volume_f16", N.ptr(Lh), N.ptr(Rh), N.ptr(vol), B, C, H, W, D, G, N.stream_ptr())
        assert ei.value.code == N.ERR_UNSUPPORTED
        close(S.build_gwc_volume(Lh, Rh, D, G), oracle.gwc_volume(L, R, D, G), **VOL_TOL)
        close(S.CostVolume(D)(Lh, Rh), oracle.corr_volume(L, R, D), **VOL_TOL)
    # mixed precisions take the fp32 path
    Lh, L = h16(rng.standard_normal((1, 8, 2, 40), dtype=np.float32))
    R = rng.standard_normal((1, 8, 2, 40), dtype=np.float32)
    close(S.CostVolume(6)(Lh, torch.from_numpy(R).cuda()), oracle.corr_volume(L, R, 6), **VOL_TOL)


@pytest.mark.parametrize("cfg", [(2, 16, 5, 40, 12, None), (1, 32, 3, 200, 192, None), (1, 48, 2, 72, 33, 20), (1, 256, 2, 312, 64, None),
                                 (1, 256, 2, 240, 48, 256), (1, 8, 1, 64, 1, 5), (1, 20, 1, 96, 32, None), (1, 24, 2, 516, 150, 24),
                                 (2, 32, 3, 960, 192, None)])
@pytest.mark.parametrize("kern", [0, 8])
def test_cross_attention_fp16_features(S, oracle, cfg, kern):
    """A7 on fp16 features: C <= 32 with value = right features -> online-softmax kernel; everything else -> the
    channel-chunked pipeline (incl. a separate value tensor, Cv != C, C = 256)."""
    B, C, H, W, D, Cv = cfg
    rng = np.random.default_rng(73)
    Lh, L = h16(rng.standard_normal((B, C, H, W), dtype=np.float32))
    Rh, R = h16(rng.standard_normal((B, C, H, W), dtype=np.float32))
    Vh, V = (None, None) if Cv is None else h16(rng.standard_normal((B, Cv, H, W), dtype=np.float32))
    for mask in ("zero", "neginf"):
        for scale in ("sqrt_c", "mean"):
            with S._native.test_options(att_kernel=kern):
                o, d = S.CrossAttentionCostVolume(D, mask, scale)(Lh, Rh, Vh)
                _, d2 = S.CrossAttentionCostVolume(D, mask, scale)(Lh, Rh, need_out=False)
            oo, od = oracle.cross_attention(L, R, V, D, mask, scale)
            close(o, oo, rtol=1e-4, atol=5e-5)
            disp_close(d, od, D)
            disp_close(d2, od, D)


@pytest.mark.parametrize("cfg", [(1, 256, 2, 240, 48), (1, 64, 3, 312, 64), (2, 40, 2, 128, 100), (1, 33, 1, 96, 20), (1, 256, 1, 516, 192)])
@pytest.mark.parametrize("waves", [0, 4, 5, 8])
@pytest.mark.parametrize("kern", [6, 8])
def test_cross_attention_chunked_kernel_fp32(S, oracle, cfg, waves, kern):
    """The channel-chunked item pipeline on fp32 features, in its fp32-MFMA form (hook 6) and its split-bf16 form (hook 8; bands
    of <= 64, longer ones fall through to the fp32 form): every workgroup shape, 1 / auto tiles per persistent workgroup, both
    masks."""
    B, C, H, W, D = cfg
    rng = np.random.default_rng(74)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32)
    R = rng.standard_normal((B, C, H, W), dtype=np.float32)
    V = rng.standard_normal((B, 40, H, W), dtype=np.float32)
    dev = lambda a: torch.from_numpy(a).cuda()
    for mask in ("zero", "neginf"):
        oo, od = oracle.cross_attention(L, R, V, D, mask, "sqrt_c")
        for tpw in (0, 1):
            with S._native.test_options(att_kernel=kern, att_waves=waves, tiles_per_wg=tpw):
                o, d = S.CrossAttentionCostVolume(D, mask, "sqrt_c")(dev(L), dev(R), dev(V))
                _, d2 = S.CrossAttentionCostVolume(D, mask, "sqrt_c")(dev(L), dev(R), need_out=False)
            close(o, oo, rtol=1e-4, atol=5e-5)
            disp_close(d, od, D)
            disp_close(d2, od, D)


@pytest.mark.parametrize("cfg", [(1, 256, 2, 312, 64), (2, 64, 3, 240, 48), (1, 40, 2, 128, 100), (1, 33, 1, 96, 20), (1, 100, 2, 516, 128),
                                 (2, 48, 5, 36, 7), (1, 17, 3, 644, 33), (1, 272, 1, 160, 96)])
@pytest.mark.parametrize("waves", [0, 4, 5, 8])
@pytest.mark.parametrize("half", [False, True])
def test_cross_attention_disparity_only_chunked16(S, oracle, cfg, waves, half):
    """A7 without a value tensor at C > 32 (default route: attention_chunked16.hip, split-fp16 operands, 16-channel items) against
    the oracle: every workgroup shape, 1 / 2 / auto tiles per persistent workgroup (item pipeline across tile and row ends), both
    masks and scales, channel counts that are no multiple of 16, bands up to 128, fp32 and fp16 features; features of magnitude
    ~30 (the split must hold fp32-level accuracy on peaked softmaxes) and of very different magnitude left and right; and
    agreement with the split-bf16 form (hook 8)."""
    B, C, H, W, D = cfg
    rng = np.random.default_rng(79)
    L = rng.standard_normal((B, C, H, W), dtype=np.float32)
    R = rng.standard_normal((B, C, H, W), dtype=np.float32)
    if half:
        (Ld, L), (Rd, R) = h16(L), h16(R)
    else:
        Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    for mask, scale in (("zero", "sqrt_c"), ("neginf", "mean"), ("neginf", "sqrt_c")):
        _, od = oracle.cross_attention(L, R, None, D, mask, scale)
        for tpw in (0, 1, 2):
            with S._native.test_options(att_waves=waves, tiles_per_wg=tpw):
                _, d = S.CrossAttentionCostVolume(D, mask, scale)(Ld, Rd, need_out=False)
            disp_close(d, od, D)
        with S._native.test_options(att_kernel=8, att_waves=waves):
            _, d8 = S.CrossAttentionCostVolume(D, mask, scale)(Ld, Rd, need_out=False)
        disp_close(d, d8.cpu().numpy(), D)
    if not half:
        L30, R30 = (L * 6).astype(np.float32), (R * 5).astype(np.float32)
        _, od = oracle.cross_attention(L30, R30, None, D, "neginf", "mean")
        with S._native.test_options(att_waves=waves):
            _, d = S.CrossAttentionCostVolume(D, "neginf", "mean")(torch.from_numpy(L30).cuda(), torch.from_numpy(R30).cuda(), need_out=False)
        disp_close(d, od, D)
        # small values on one side, large ones on the other (the low fp16 piece of a value below 2^-7 is subnormal: absolute, not
        # relative, accuracy -- the scores must not notice)
        Ls, Rb = (L * 0.004).astype(np.float32), (R * 250).astype(np.float32)
        _, od = oracle.cross_attention(Ls, Rb, None, D, "neginf", "sqrt_c")
        with S._native.test_options(att_waves=waves):
            _, d = S.CrossAttentionCostVolume(D, "neginf", "sqrt_c")(torch.from_numpy(Ls).cuda(), torch.from_numpy(Rb).cuda(), need_out=False)
        disp_close(d, od, D)


def test_cross_attention_disparity_only_chunked16_random_shapes(S, oracle):
    """48 seeded random shapes of the disparity-only C > 32 route (C 33..300, W 36..900 in steps of 4, D 1..128, 1..3 rows, 1..2
    pairs, fp32 / fp16 features, both masks, random tiles per workgroup) against the oracle."""
    rng = np.random.default_rng(2025)
    for i in range(48):
        B, C, H = int(rng.integers(1, 3)), int(rng.integers(33, 301)), int(rng.integers(1, 4))
        W, D = 4 * int(rng.integers(9, 226)), int(rng.integers(1, 129))
        sr = float(rng.choice([0.05, 1.0, 8.0]))
        sl = min(float(rng.choice([0.05, 1.0, 8.0])), 2.0 / sr)   # logits of a few hundred make the 1e-3 bar a test of fp32 itself
        L = rng.standard_normal((B, C, H, W), dtype=np.float32) * sl
        R = rng.standard_normal((B, C, H, W), dtype=np.float32) * sr
        mask, scale = ("zero", "neginf")[i & 1], ("sqrt_c", "mean")[(i >> 1) & 1]
        if i % 3 == 2:
            (Ld, L), (Rd, R) = h16(L), h16(R)
        else:
            Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        _, od = oracle.cross_attention(L, R, None, D, mask, scale)
        with S._native.test_options(tiles_per_wg=int(rng.integers(0, 4)), att_waves=int(rng.choice([0, 0, 4, 5, 8]))):
            _, d = S.CrossAttentionCostVolume(D, mask, scale)(Ld, Rd, need_out=False)
        assert torch.isfinite(d).all(), (i, B, C, H, W, D)
        disp_close(d, od, D)


@pytest.mark.parametrize("shape", [(1, 256, 3, 240), (2, 32, 2, 152), (1, 64, 1, 36), (1, 40, 2, 312), (1, 8, 1, 128)])
def test_allpairs_fp16_features(S, oracle, shape):
    """A8 on fp16 feature maps (what the fnet hands over under autocast; the reference widens with .float()): pyramid == oracle
    on the rounded values, and the lookup on top of it."""
    B, C, H, W = shape
    rng = np.random.default_rng(75)
    f1h, f1 = h16(rng.standard_normal(shape, dtype=np.float32))
    f2h, f2 = h16(rng.standard_normal(shape, dtype=np.float32))
    blk = S.CorrBlock1D(f1h, f2h, num_levels=4, radius=4)
    pyr = oracle.corr_pyramid(oracle.allpairs_corr(f1, f2), 4)
    for lv, ref in zip(blk.corr_pyramid, pyr):
        close(lv.reshape(B * H * W, -1), ref, **VOL_TOL)
    coords = S.coords_grid(B, H, W, device="cuda") - torch.rand(B, 2, H, W, device="cuda") * 9
    close(blk(coords), oracle.corr_lookup(pyr, coords.cpu().numpy(), 4, 4), rtol=1e-4, atol=5e-5)
    close(S.CorrBlock1D.corr(f1h, f2h).reshape(B, H, W, W), oracle.allpairs_corr(f1, f2), **VOL_TOL)
    # identical to widening first (what the reference does): bit for bit
    blk32 = S.CorrBlock1D(f1h.float(), f2h.float(), num_levels=4, radius=4)
    for a, b in zip(blk.corr_pyramid, blk32.corr_pyramid):
        assert torch.equal(a, b)


def test_instance_norm_relu(S, oracle, golden):
    """Fused InstanceNorm2d + ReLU epilogue: fp32 output against the tensors captured inside the reference's extractor
    (golden G14) and against the oracle; fp16 output == the fp32 result rounded once (<= 1 fp16 ulp)."""
    g = golden("g14_extractor_epilogue")
    dev = lambda a: torch.from_numpy(a).cuda()
    for x, y in ((g["x1"], g["y1"]), (g["x2"], g["y2"])):
        o32 = S.instance_norm_relu(dev(x), out_dtype=torch.float32)
        close(o32, y, rtol=1e-5, atol=3e-6)
        o16 = S.instance_norm_relu(dev(x))
        assert o16.dtype == torch.float16 and o16.is_contiguous()
        assert torch.equal(o16, o32.half())                     # one rounding, no double rounding
    m = S.InstanceNormReLU(6, eps=1e-3, affine=True, relu=False, out_dtype=torch.float32).cuda()
    m.weight.data, m.bias.data = dev(g["wa"]), dev(g["ba"])
    close(m(dev(g["xa"])), g["ya"], rtol=1e-5, atol=5e-6)
    assert set(m.state_dict().keys()) == {"weight", "bias"}     # nn.InstanceNorm2d(affine=True) keys
    # larger planes (several chunks per plane, ragged last chunk), large mean / small spread
    rng = np.random.default_rng(76)
    x = (rng.standard_normal((2, 5, 136, 244), dtype=np.float32) * 0.5 + 7.0).astype(np.float32)
    close(S.instance_norm_relu(dev(x), out_dtype=torch.float32), oracle.instance_norm_relu(x), rtol=1e-4, atol=2e-5)
    with pytest.raises(S._native.StereoHipError, match="multiple of 4"):
        S.instance_norm_relu(torch.randn(1, 2, 3, 5, device="cuda"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        S.instance_norm_relu(torch.randn(1, 2, 4, 4))


def test_cfg2_chain_fp16_handover_full_size_rows(S, oracle):
    """cfg-2 geometry (960 wide, D=192, C=32, G=8), extractor epilogue -> fp16 features -> fused volume stage -> local lookup,
    against the oracle chain fed the same rounded features: disparity within 1e-3; and the fp16 rounding of the features
    itself moves the disparity by far less than the bar on matched (shifted) pairs."""
    rng = np.random.default_rng(77)
    B, C, H, W, D, G = 1, 32, 6, 960, 192, 8
    raw = rng.standard_normal((B, C, H, W), dtype=np.float32)
    rawR = np.roll(raw, -17, axis=3) + 0.05 * rng.standard_normal((B, C, H, W), dtype=np.float32)
    dev = lambda a: torch.from_numpy(a).cuda()
    Lh, Rh = S.instance_norm_relu(dev(raw)), S.instance_norm_relu(dev(rawR))          # fp16 NCHW features
    L, R = Lh.float().cpu().numpy(), Rh.float().cpu().numpy()
    volg, disp, mean = S.build_gwc_volume_with_disparity(Lh, Rh, D, G, return_mean=True)
    refined = S.LocalCostVolume(2, 20)(mean, disp.unsqueeze(1), True)
    ovol = oracle.gwc_volume(L, R, D, G)
    omean = oracle.group_mean(ovol)
    odisp = oracle.soft_argmin(omean)
    close(volg, ovol, **VOL_TOL)
    disp_close(disp, odisp, D)
    disp_close(refined, oracle.local_lookup(omean, odisp[:, None], 2.0, 20, True), D)
    # effect of the fp16 hand-over itself (fp32 features of the same epilogue vs their fp16 rounding)
    L32, R32 = S.instance_norm_relu(dev(raw), out_dtype=torch.float32), S.instance_norm_relu(dev(rawR), out_dtype=torch.float32)
    _, disp32 = S.build_gwc_volume_with_disparity(L32, R32, D, G)
    assert (disp32 - disp).abs().max().item() < 5e-2     # informational bound: fp16 features, not a parity claim


@pytest.mark.parametrize("shape", [(2, 256, 5, 120, 24), (1, 64, 3, 300, 64), (1, 40, 2, 516, 100), (1, 33, 2, 128, 128), (2, 96, 1, 36, 7),
                                   (1, 256, 2, 960, 24), (1, 48, 3, 8, 5)])
@pytest.mark.parametrize("half", [False, True])
def test_corr_volume_channel_chunked(S, oracle, shape, half):
    """A1 and A1+A4 for C > 32 on the matrix cores (32-channel chunks through two LDS buffers, band tiles accumulated in
    registers, skew sheet, row-contiguous flush) -- incl. the reference-native shape class (256 channels, D = 24,
    baseline.py:63,114) -- against the oracle and against the VALU fallback; fp32 and fp16 features."""
    B, C, H, W, D = shape
    rng = np.random.default_rng(81)
    L0 = rng.standard_normal((B, C, H, W), dtype=np.float32)
    R0 = rng.standard_normal((B, C, H, W), dtype=np.float32)
    if half:
        Ld, L = h16(L0)
        Rd, R = h16(R0)
    else:
        Ld, Rd, L, R = torch.from_numpy(L0).cuda(), torch.from_numpy(R0).cuda(), L0, R0
    ref = oracle.corr_volume(L, R, D)
    close(S.CostVolume(D)(Ld, Rd), ref, **VOL_TOL)
    for sim in (True, False):
        for tpw in (0, 1, 3):
            with S._native.test_options(tiles_per_wg=tpw):
                v, d = S.corr_volume_with_disparity(Ld, Rd, D, match_similarity=sim)
            close(v, ref, **VOL_TOL)
            disp_close(d, oracle.soft_argmin(ref, sim), D)
    with S._native.test_options(corr_mfma=0):        # the VALU kernel it replaces at these shapes
        close(S.CostVolume(D)(Ld.float(), Rd.float()), ref, **VOL_TOL)


@pytest.mark.parametrize("shape", [(2, 32, 12, 40), (1, 64, 135, 244), (2, 128, 9, 36), (1, 256, 5, 16), (1, 4, 3, 8), (1, 48, 6, 10)])
def test_instance_norm_relu_channels_last_input(S, oracle, shape):
    """N2, channels-last hand-over: a torch.channels_last activation (fp32 or fp16) is normalised and written NCHW in one op
    (no .contiguous() pass): equal to the oracle on the same values; equal to the NCHW kernel's fp16 output up to the last
    fp32 bit of the statistics (<= 1 fp16 ulp); affine, no-ReLU and the unsupported-C fallback."""
    from stereoformer_amd import extractor_epilogue as E
    rng = np.random.default_rng(77)
    B, C, H, W = shape
    x = (rng.standard_normal(shape, dtype=np.float32) * 1.5 + 0.7).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    xcl = xt.contiguous(memory_format=torch.channels_last)
    assert E._channels_last_operand(xcl) == (C % 4 == 0 and 256 % (C // 4) == 0 and not xcl.is_contiguous())
    ref = oracle.instance_norm_relu(x)
    o32 = S.instance_norm_relu(xcl, out_dtype=torch.float32)
    assert o32.is_contiguous() and o32.shape == xt.shape
    close(o32, ref, rtol=1e-4, atol=2e-5)
    o16 = S.instance_norm_relu(xcl)
    assert o16.dtype == torch.float16 and o16.is_contiguous()
    close(o16.float(), ref, rtol=2e-3, atol=1e-3)
    assert (o16.float() - S.instance_norm_relu(xt).float()).abs().max().item() <= 2e-3   # vs the NCHW kernel
    # fp16 activations (autocast extractor): the oracle on the rounded values
    xh = xcl.half()
    assert xh.is_contiguous(memory_format=torch.channels_last)
    close(S.instance_norm_relu(xh, out_dtype=torch.float32), oracle.instance_norm_relu(xh.float().cpu().numpy()), rtol=1e-4, atol=2e-5)
    # affine, no ReLU
    g = rng.standard_normal(C).astype(np.float32)
    bt = rng.standard_normal(C).astype(np.float32)
    close(S.instance_norm_relu(xcl, torch.from_numpy(g).cuda(), torch.from_numpy(bt).cuda(), eps=1e-3, relu=False, out_dtype=torch.float32),
          oracle.instance_norm_relu(x, g, bt, eps=1e-3, relu=False), rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_channels_last_features_take_the_relayout_kernel(S, dtype):
    """Channels-last feature tensors (an extractor running in torch.channels_last that ends in a plain convolution) are brought
    to NCHW by the library's relayout kernel: bit-identical to the NCHW tensor, and every feature-consuming surface gives the
    same numbers for both layouts; shapes outside the kernel's envelope (C % 4 != 0, ragged H*W) take contiguous()."""
    from stereoformer_amd import _native as N
    torch.manual_seed(78)
    for shape in ((2, 32, 6, 40), (1, 256, 5, 16), (1, 8, 3, 12), (1, 6, 3, 12), (1, 8, 3, 5)):
        x = torch.randn(*shape, device="cuda").to(dtype)
        xcl = x.contiguous(memory_format=torch.channels_last)
        y = N.nchw(xcl)
        assert y.is_contiguous() and y.dtype == dtype and torch.equal(y, x)
        assert N.nchw(x) is x
    L = torch.randn(2, 32, 6, 40, device="cuda").to(dtype)
    R = torch.randn(2, 32, 6, 40, device="cuda").to(dtype)
    Lc, Rc = (t.contiguous(memory_format=torch.channels_last) for t in (L, R))
    with torch.no_grad():
        assert torch.equal(S.CostVolume(12)(Lc, Rc), S.CostVolume(12)(L, R))
        a, b = S.build_gwc_volume_with_disparity(Lc, Rc, 12, 8), S.build_gwc_volume_with_disparity(L, R, 12, 8)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        a, b = S.CrossAttentionCostVolume(12)(Lc, Rc), S.CrossAttentionCostVolume(12)(L, R)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for la, lb in zip(S.CorrBlock1D(Lc, Rc, 2, 2).corr_pyramid, S.CorrBlock1D(L, R, 2, 2).corr_pyramid):
            assert torch.equal(la, lb)
