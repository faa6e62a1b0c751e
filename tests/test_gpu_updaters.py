"""GPU parity of the GRU updaters (A11, A14) against golden tensors from the real reference
(3 / 2 iterations with RNG-free deterministic weights, tests/golden/_weights.py).  The
convolutions run in MIOpen (host framework) so the tolerance is the conv stack's fp32 noise:
hidden states / masks 2e-4, disparities 1e-3 (the north-star bar)."""
import numpy as np
import pytest
import torch

from _weights import fill_state_dict_

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def close(a, b, **tol):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b, **tol)


@pytest.fixture(scope="module")
def S():
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    import stereoformer_amd as s
    return s


def test_disparity_update_dlc_with_mask_g5(S, golden):
    g = golden("g5_gru_updater")
    D = g["vol"].shape[1]
    net = S.DisparityUpdateDLCWithMask(input_channels=D, hidden_dim=8, sample_points=20).eval()
    fill_state_dict_(net, seed=0.5)
    net = net.cuda()
    vol, disp, left, right = dev(g["vol"]), dev(g["disp0"]), dev(g["left"]), dev(g["right"])
    # pieces first
    h, w = disp.shape[2:]
    close(S.uncertainty_volume(vol, disp), g["uncertainty0"], rtol=1e-4, atol=2e-4)
    li = S.resize_bilinear(left, (h, w), False)
    ri = S.resize_bilinear(right, (h, w), False)
    close(li, g["left_small"], rtol=1e-5, atol=1e-5)
    close(S.disp_warp(ri, disp)[0] - li, g["error0"], rtol=1e-5, atol=2e-5)
    hidden = None   # the reference needs an explicit zero hidden on CPU; None -> zeros here
    for it in range(3):
        disp, hidden, mask = net(vol, disp, left, right, hidden, True)
        assert mask.shape == (1, 576, h, w)
        close(hidden, g[f"hidden{it + 1}"], rtol=1e-4, atol=2e-4)
        close(mask[:, ::37], g[f"mask{it + 1}_sub"], rtol=1e-4, atol=2e-4)
        close(disp, g[f"disp{it + 1}"], rtol=0, atol=1e-3)


def test_basic_multi_update_block_g8(S, golden):
    g = golden("g8_raft_update")
    blk = S.BasicMultiUpdateBlock_LZH(4, 4, 3, 2, [16, 16, 16]).eval()
    fill_state_dict_(blk, seed=0.25)
    blk = blk.cuda()
    net = [dev(g[f"net{i}_0"]) for i in range(3)]
    inp = [[dev(g[f"inp{i}{j}"]) for j in range(3)] for i in range(3)]
    corr, flow = dev(g["corr"]), dev(g["flow"])
    for it in range(2):
        net, mask, delta = blk(net, inp, corr, flow, iter32=True, iter16=True)
        for i in range(3):
            close(net[i], g[f"net{i}_{it + 1}"], rtol=1e-4, atol=2e-4)
        close(mask[:, ::7], g[f"mask_{it + 1}"], rtol=1e-4, atol=2e-4)
        close(delta, g[f"delta_{it + 1}"], rtol=1e-4, atol=2e-4)
        flow = flow + delta


def test_sepconvgru_and_softsign_run(S):
    """Smoke for the remaining reference surfaces (no golden: unused by the reference models)."""
    gru = S.SepConvGRU(hidden_dim=8, input_dim=12).cuda().eval()
    h = torch.tanh(torch.randn(1, 8, 6, 10, device="cuda"))
    x = torch.randn(1, 12, 6, 10, device="cuda")
    with torch.no_grad():
        out = gru(h, x)
        # same math in plain torch
        ref = h
        for cz, cr, cq in ((gru.convz1, gru.convr1, gru.convq1), (gru.convz2, gru.convr2, gru.convq2)):
            hx = torch.cat([ref, x], 1)
            z, r = torch.sigmoid(cz(hx)), torch.sigmoid(cr(hx))
            q = torch.tanh(cq(torch.cat([r * ref, x], 1)))
            ref = (1 - z) * ref + z * q
    assert (out - ref).abs().max().item() < 1e-5
    from stereoformer_amd.update_dlc import ConvGRU
    g2 = ConvGRU(4, softsign=True).cuda().eval()
    inp = torch.randn(1, 4, 5, 7, device="cuda")
    with torch.no_grad():
        o = g2(inp, None)
        x1 = torch.cat((inp, torch.zeros_like(inp)), 1)
        z = g2.conv_zz(x1)
        b = g2.conv_bb(x1)
        gg = g2.conv_gg(torch.cat((b * 0, inp), 1))
        ref = z * gg
    assert (o - ref).abs().max().item() < 1e-5


def test_lowcnn_gru_loop_16_iters_no_sync(S):
    """cfg-4 LowCNN-GRU leg at reduced batch: 16 x (A11 + A13 f=8) on a 1/8-scale volume;
    checks shapes, finiteness and that disparities stay inside the valid range semantics."""
    B, D, h, w = 2, 24, 72, 120
    torch.manual_seed(0)
    net = S.DisparityUpdateDLCWithMask(D, 32, 20).eval().cuda()
    vol = torch.randn(B, D, h, w, device="cuda")
    left = torch.randn(B, 3, 8 * h, 8 * w, device="cuda")
    right = torch.randn(B, 3, 8 * h, 8 * w, device="cuda")
    disp = S.DisparityEstimation(D)(vol).unsqueeze(1)
    hidden = None
    for _ in range(16):
        disp, hidden, mask = net(vol, disp, left, right, hidden, True)
        up = S.upsample_convex8(disp, mask)
    assert up.shape == (B, 1, 8 * h, 8 * w) and torch.isfinite(up).all()
    assert disp.min().item() >= 0 and disp.max().item() <= D


def _raft_inputs(g):
    net = [dev(g[f"net{i}"]) for i in range(3)]
    inp = [[dev(g[f"inp{i}{j}"]) for j in range(3)] for i in range(3)]
    return dev(g["fmap1"]), dev(g["fmap2"]), net, inp


def test_raft_stereo_loop_end_to_end_g9(S, golden):
    """cfg-4 RAFT leg: the re-hosted loop (A8 -> 4 x (A9 + A14 + coords) -> A13) fed with the tensors the
    reference's extractors produced, against the reference's own final outputs."""
    g = golden("g9_raft_loop")
    loop = S.RAFTStereoLoop(hidden_dims=[32, 32, 32]).eval()
    # same RNG-free recipe as on the reference side, where it was applied to the WHOLE RAFT_Stereo model:
    # the fixture carries that model's sorted key list so the update block gets identical values
    fill_state_dict_(loop.update_block, seed=0.125, keys=list(g["keys"]), prefix="update_block.")
    loop = loop.cuda()
    for p_ in loop.parameters():
        p_.requires_grad_(False)
    f1, f2, net, inp = _raft_inputs(g)
    flow_lo, flow_up = loop(f1, f2, net, inp, iters=4, test_mode=True)
    close(flow_lo, g["flow_lo"], rtol=1e-4, atol=1e-3)
    close(flow_up, g["flow_up"], rtol=1e-4, atol=2e-3)      # x4 upsampled flow: 4x the 1/4-scale error
    assert float(np.abs(flow_up.cpu().numpy() - g["flow_up"]).mean()) < 2e-4
    f1, f2, net, inp = _raft_inputs(g)
    preds = loop(f1, f2, net, inp, iters=3, test_mode=False)
    assert len(preds) == 3
    for k, p in enumerate(preds):
        close(p, g[f"pred{k}"], rtol=1e-4, atol=2e-3)
    # HIP-graph replay of the iteration == eager
    f1, f2, net, inp = _raft_inputs(g)
    with torch.no_grad():
        lo_g, up_g = loop(f1, f2, net, inp, iters=4, test_mode=True, use_graph=True)
    # flows are O(10): a few ulp (MIOpen may pick another conv algorithm in the warm-up before capture)
    assert (lo_g - flow_lo).abs().max().item() < 5e-5 and (up_g - flow_up).abs().max().item() < 1e-4


def test_raft_loop_channels_last_matches_reference(S, golden, monkeypatch):
    """NHWC weights / activations through the update block -- whose 3x3 convolutions then run on the split-bf16 matrix-core
    kernel (csrc/conv_split.hip; asserted) -- must still reproduce the golden outputs of the reference RAFT_Stereo (G9) at the
    fp32 parity tolerances."""
    from stereoformer_amd import conv_split as C
    calls = []
    real = C.conv_nhwc
    monkeypatch.setattr(C, "conv_nhwc", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    g = golden("g9_raft_loop")
    loop = S.RAFTStereoLoop(hidden_dims=[32, 32, 32], channels_last=True).eval()
    fill_state_dict_(loop.update_block, seed=0.125, keys=list(g["keys"]), prefix="update_block.")
    loop = loop.cuda()
    with torch.no_grad():
        f1, f2, net, inp = _raft_inputs(g)
        flow_lo, flow_up = loop(f1, f2, net, inp, iters=4, test_mode=True)
        close(flow_lo, g["flow_lo"], rtol=1e-4, atol=1e-3)
        close(flow_up, g["flow_up"], rtol=1e-4, atol=2e-3)
        f1, f2, net, inp = _raft_inputs(g)
        assert len(calls) >= 4 * (3 * 2 + 3 + 1)     # per iteration: 3 GRUs x (z | r, q), 3 encoder convolutions, the head(s)
        lo_g, up_g = loop(f1, f2, net, inp, iters=4, test_mode=True, use_graph=True)
        assert (up_g - flow_up).abs().max().item() < 1e-4
    assert flow_up.is_contiguous()


def test_raft_loop_mixed_precision_option(S, golden):
    """The reference's optional ``mixed_precision`` flag (update-block convolutions under fp16 autocast, everything
    else fp32): not a parity configuration -- checked to stay close to the fp32 result of the same loop."""
    g = golden("g9_raft_loop")
    loop = S.RAFTStereoLoop(hidden_dims=[32, 32, 32]).eval()
    fill_state_dict_(loop.update_block, seed=0.125, keys=list(g["keys"]), prefix="update_block.")
    loop = loop.cuda()
    with torch.no_grad():
        f1, f2, net, inp = _raft_inputs(g)
        lo32, up32 = loop(f1, f2, net, inp, iters=4, test_mode=True)
        loop.mixed_precision = True
        f1, f2, net, inp = _raft_inputs(g)
        lo16, up16 = loop(f1, f2, net, inp, iters=4, test_mode=True)
        f1, f2, net, inp = _raft_inputs(g)
        lo16g, up16g = loop(f1, f2, net, inp, iters=4, test_mode=True, use_graph=True)
    assert lo16.dtype == torch.float32 and up16.dtype == torch.float32
    assert (up16 - up32).abs().mean().item() < 0.05 and (up16 - up32).abs().max().item() < 1.0
    # fp16 convolutions: MIOpen may pick another algorithm for the captured replay -> fp16-level differences
    assert (up16g - up16).abs().mean().item() < 0.02 and (up16g - up16).abs().max().item() < 0.5


def test_lowcnn_gru_loop_graph_matches_eager(S):
    B, D, h, w = 2, 24, 36, 60
    torch.manual_seed(1)
    loop = S.LowCNNGRULoop(D, 32, 20).eval().cuda()
    vol = torch.randn(B, D, h, w, device="cuda")
    left = torch.randn(B, 3, 8 * h, 8 * w, device="cuda")
    right = torch.randn(B, 3, 8 * h, 8 * w, device="cuda")
    disp = S.DisparityEstimation(D)(vol).unsqueeze(1)
    with torch.no_grad():
        eager = loop(vol, disp, left, right, iters=6)
        graphed = loop(vol, disp, left, right, iters=6, use_graph=True)
        graphed_all = loop(vol, disp, left, right, iters=6, use_graph=True, graph_all_iterates=True)
    assert len(graphed_all) == 6
    for a_, b_ in zip(eager, graphed_all):
        assert (a_ - b_).abs().max().item() < 1e-4
    with pytest.raises(RuntimeError, match="inference path"):
        loop(vol, disp, left, right, iters=6, use_graph=True)
    # and the same loop is trainable in grad mode
    outs = loop(vol, disp, left, right, iters=2)
    outs[-1].mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in loop.parameters())
    assert len(eager) == 6 and len(graphed) == 1
    assert (eager[-1] - graphed[0]).abs().max().item() < 1e-4


def test_raft_loop_flow_init_g13(S, golden):
    """The loop started from a given disparity field (`flow_init`, raft_stereo.py:163-164) against the reference
    RAFT_Stereo run the same way (golden G13), eager and as a HIP-graph replay."""
    g = golden("g13_flow_init_padder")
    loop = S.RAFTStereoLoop(hidden_dims=[32, 32, 32]).eval()
    fill_state_dict_(loop.update_block, seed=0.125, keys=list(golden("g9_raft_loop")["keys"]), prefix="update_block.")
    loop = loop.cuda()
    with torch.no_grad():
        f1, f2, net, inp = _raft_inputs(g)
        flow_lo, flow_up = loop(f1, f2, net, inp, iters=3, flow_init=dev(g["flow_init"]), test_mode=True)
        close(flow_lo, g["flow_lo"], rtol=1e-4, atol=1e-3)
        close(flow_up, g["flow_up"], rtol=1e-4, atol=2e-3)
        assert float(np.abs(flow_up.cpu().numpy() - g["flow_up"]).mean()) < 2e-4
        f1, f2, net, inp = _raft_inputs(g)
        lo_g, up_g = loop(f1, f2, net, inp, iters=3, flow_init=dev(g["flow_init"]), test_mode=True, use_graph=True)
    assert (lo_g - flow_lo).abs().max().item() < 5e-5 and (up_g - flow_up).abs().max().item() < 1e-4
    # and without flow_init the result differs (the argument is not ignored)
    with torch.no_grad():
        f1, f2, net, inp = _raft_inputs(g)
        lo0, _ = loop(f1, f2, net, inp, iters=3, test_mode=True)
    assert (lo0 - flow_lo).abs().max().item() > 1e-2


def _loop_inputs(B, Hq, Wq, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=gen)
    f1 = rn(B, 256, Hq, Wq)
    f2 = torch.roll(f1, 5, 3) + 0.1 * rn(B, 256, Hq, Wq)
    nets = [torch.tanh(rn(B, 128, Hq >> i, Wq >> i)) for i in range(3)]
    inps = [[rn(B, 128, Hq >> i, Wq >> i) for _ in range(3)] for i in range(3)]
    return f1, f2, nets, inps


def _check_final_iterate(S, oracle, loop, f1, f2, nets, inps, iters, flow_init=None):
    """Runs the loop eagerly and as a graph replay; checks graph == eager, and the custom ops of the LAST iteration against
    the oracle fed the loop's own tensors: the pyramid lookup at the final coordinates and the convex upsampling of the
    final flow with the final mask."""
    cap = {}

    def hook(mod, args, kwargs, out):
        cap["corr"], cap["flow"], cap["mask"] = args[2], args[3], out[1]

    hk = loop.update_block.register_forward_hook(hook, with_kwargs=True)
    with torch.no_grad():
        flow_lo, flow_up = loop(f1, f2, [n.clone() for n in nets], inps, iters=iters, flow_init=flow_init, test_mode=True)
    hk.remove()
    assert torch.isfinite(flow_lo).all() and torch.isfinite(flow_up).all()
    assert float(flow_lo[:, 1].abs().max()) == 0.0               # raft_stereo.py:180: the y component never moves
    with torch.no_grad():
        lo_g, up_g = loop(f1, f2, [n.clone() for n in nets], inps, iters=iters, flow_init=flow_init, test_mode=True,
                          use_graph=True)
    # 16 iterations amplify the ulp-level differences of MIOpen's algorithm choice in the graph's warm-up
    assert (lo_g - flow_lo).abs().max().item() < 2e-3 and (up_g - flow_up).abs().max().item() < 8e-3
    B, _, Hq, Wq = f1.shape
    # A8 + A9 at the coordinates the last update block saw: coords0 + flow fed to it
    coords0 = S.coords_grid(B, Hq, Wq, device="cuda")
    coords_last = (coords0 + cap["flow"].float()).cpu().numpy()
    pyr = oracle.corr_pyramid(oracle.allpairs_corr(f1.cpu().numpy(), f2.cpu().numpy()), 4)
    ref_corr = oracle.corr_lookup(pyr, coords_last, 4, 4)
    # (the inference HIP path hands the block the 36 features channels-last, zero-padded to 48 channels)
    got = cap["corr"].float()
    assert got.shape[1] in (36, 48) and float(got[:, 36:].abs().sum()) == 0.0
    close(got[:, :36].contiguous(), ref_corr, rtol=1e-4, atol=2e-4)
    # A13 on the final iterate
    ref_up = oracle.upsample_convex(flow_lo.cpu().numpy(), cap["mask"].float().cpu().numpy(), 4)[:, :1]
    close(flow_up, ref_up, rtol=1e-5, atol=1e-4)
    return flow_lo, flow_up


def test_raft_loop_16_iterations_full_size(S, oracle):
    """cfg-4 per-GPU geometry, two pairs: 16 iterations at [2,256,136,240] (544x960 / 4): graph == eager, final lookup and
    final convex upsampling against the oracle."""
    torch.manual_seed(4)
    loop = S.RAFTStereoLoop([128, 128, 128], channels_last=True).eval().cuda()
    f1, f2, nets, inps = _loop_inputs(2, 136, 240, 41)
    _check_final_iterate(S, oracle, loop, f1, f2, nets, inps, 16)


def test_cfg5_attention_flow_init_loop_chain(S, oracle):
    """cfg-5 chain at one full-height pair: cross-attention at 1/4 resolution [1,256,96,312] (band 64, mask 'neginf') against
    the oracle -> initial flow (-disp, 0) -> 16-iteration loop: graph == eager, final lookup / upsampling against the oracle."""
    torch.manual_seed(5)
    loop = S.RAFTStereoLoop([128, 128, 128], channels_last=True).eval().cuda()
    f1, f2, nets, inps = _loop_inputs(1, 96, 312, 51)
    f1, f2 = f1 * 0.25, f2 * 0.25                                # logits O(1) at C = 256 / sqrt(C) scaling
    _, d = S.CrossAttentionCostVolume(64, "neginf", "sqrt_c")(f1, f2, need_out=False)
    _, od = oracle.cross_attention(f1.cpu().numpy(), f2.cpu().numpy(), None, 64, "neginf", "sqrt_c")
    close(d, od, rtol=0, atol=1e-3)
    assert float(np.abs(d.cpu().numpy() - od).mean()) < 2.5e-4
    init = torch.cat([-d.unsqueeze(1), torch.zeros_like(d.unsqueeze(1))], 1)
    lo, _ = _check_final_iterate(S, oracle, loop, f1, f2, nets, inps, 16, flow_init=init)
    assert (lo[:, :1] + d.unsqueeze(1)).abs().mean().item() < 50   # the loop moved from, not ignored, its start


def test_lowcnn_gru_loop_channels_last_takes_the_split_convolutions(S, monkeypatch):
    """LowCNNGRULoop in NHWC (the inference default): the updater's 3x3 convolutions (guidance encoders with their BatchNorm
    folded in, GRU gates with the gate math in the epilogues, offset head, mask head incl. its 1x1 projection) run on conv_split
    (asserted); same
    disparities as the NCHW / library-convolution loop to fp32 rounding, eager and graph."""
    import copy
    from stereoformer_amd import conv_split as C
    B, D, h, w = 2, 24, 36, 60
    torch.manual_seed(2)
    ref = S.LowCNNGRULoop(D, 32, 20, channels_last=False).eval().cuda()
    cl = S.LowCNNGRULoop(D, 32, 20).eval().cuda()
    cl.load_state_dict(copy.deepcopy(ref.state_dict()))
    vol = torch.randn(B, D, h, w, device="cuda")
    left = torch.randn(B, 3, 8 * h, 8 * w, device="cuda")
    right = torch.randn(B, 3, 8 * h, 8 * w, device="cuda")
    disp = S.DisparityEstimation(D)(vol).unsqueeze(1)
    calls = []
    real = C.conv_nhwc
    monkeypatch.setattr(C, "conv_nhwc", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        a = ref(vol, disp, left, right, iters=5)
        assert not calls
        b = cl(vol, disp, left, right, iters=5)
        assert len(calls) == 5 * 7                      # per iteration: 2 encoders, z | b and g gates, offset.conv1, mask[0], mask[2]
        g = cl(vol, disp, left, right, iters=5, use_graph=True, graph_all_iterates=True)
    for x, y, z in zip(a, b, g):
        assert x.shape == y.shape and float((x - y).abs().max()) < 2e-3 * max(1.0, float(x.abs().max()) / 50), float((x - y).abs().max())
        assert float((y - z).abs().max()) < 1e-4


def test_captured_raft_loop_does_not_keep_a_motion_buffer_per_iteration(S):
    """inside a HIP-graph capture the side-stream motion tensor must not be record_stream()ed (that defers every
    iteration's free to the end of the capture: one [B,128,h,w] buffer per iteration in the graph's private pool).  The
    pool of a 12-iteration capture may exceed that of a 4-iteration capture by far less than 8 such buffers."""
    B, Hq, Wq = 2, 64, 96
    motion_bytes = B * 128 * Hq * Wq * 4

    def graph_pool_bytes(iters):
        torch.manual_seed(3)
        loop = S.RAFTStereoLoop([128, 128, 128]).eval().cuda()
        f1, f2, nets, inps = _loop_inputs(B, Hq, Wq, 77)
        with torch.no_grad():
            loop(f1, f2, [n.clone() for n in nets], inps, iters=iters, test_mode=True)     # eager: warms every cache
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            before = torch.cuda.memory_reserved()
            loop(f1, f2, [n.clone() for n in nets], inps, iters=iters, test_mode=True, use_graph=True)
            torch.cuda.synchronize()
            grown = torch.cuda.memory_reserved() - before
        del loop
        torch.cuda.empty_cache()
        return grown

    small, large = graph_pool_bytes(4), graph_pool_bytes(12)
    assert large - small < 3 * motion_bytes, (small, large, motion_bytes)
