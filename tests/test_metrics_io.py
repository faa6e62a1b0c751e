"""metrics and the PFM container against values produced by the reference's own functions
(golden G10).  CPU tests; the disparity rescale (HIP resize kernel) has its GPU test below."""
import os

import numpy as np
import pytest
import torch


def test_metrics_match_reference(golden):
    import stereoformer_amd as S
    g = golden("g10_metrics_io")
    gt, pred = torch.from_numpy(g["gt"]), torch.from_numpy(g["pred"])
    assert abs(float(S.EPE_Loss(pred, gt)) - float(g["epe"])) < 1e-6
    assert abs(float(S.sequence_loss([pred + 0.5 * i for i in range(3)], gt)) - float(g["seq"])) < 1e-5
    assert float(S.D1_metric(pred, gt)) == float(g["d1"])
    assert float(S.P1_metric(pred, gt)) == float(g["p1"])
    assert float(S.thres_metric(pred, gt, (gt > 0) & (gt < 192), 3.0)) == float(g["thres3"])


def test_pfm_round_trip_and_reference_bytes(golden, tmp_path):
    import stereoformer_amd as S
    g = golden("g10_metrics_io")
    # a file written by the reference's writer parses to the same array the reference's reader returned
    ref_file = tmp_path / "ref.pfm"
    ref_file.write_bytes(g["pfm_bytes"].tobytes())
    data, scale = S.read_pfm(str(ref_file))
    assert scale == 1.0 and np.array_equal(data, g["pfm_back"]) and np.array_equal(data, g["pfm_img"])
    # our writer produces the same bytes as the reference's
    mine = tmp_path / "mine.pfm"
    S.write_pfm(str(mine), g["pfm_img"])
    assert mine.read_bytes() == g["pfm_bytes"].tobytes()
    assert np.array_equal(S.read_disp(str(mine)), g["pfm_img"])
    assert np.array_equal(S.read_disp(str(mine), subset=True), -g["pfm_img"])
    with pytest.raises(Exception):
        S.write_pfm(str(mine), g["pfm_img"].astype(np.float64))
    with pytest.raises(Exception):
        S.read_disp("x.png")


@pytest.mark.gpu
def test_scale_disp_gpu(golden):
    import stereoformer_amd as S
    g = golden("g10_metrics_io")
    out = S.scale_disp(torch.from_numpy(g["scale_in"]).cuda(), (1, 9, 16))
    np.testing.assert_allclose(out.cpu().numpy(), g["scale_out"], rtol=1e-5, atol=1e-5)
