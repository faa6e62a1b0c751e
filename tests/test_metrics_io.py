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


def test_stereo_dataset_and_transforms_golden(golden, tmp_path):
    """StereoDataset + RandomCrop / ToTensor / Normalize against the reference's own loader run on the same tiny tree
    (golden G16: the fixture carries the source arrays, the files are re-created here), incl. the saved left name and the
    KITTI 16-bit PNG disparity reader."""
    from PIL import Image

    import stereoformer_amd as S
    from stereoformer_amd import data as T
    g = golden("g16_dataset")
    lines = []
    for i in range(2):
        d = tmp_path / "frames" / str(i)
        d.mkdir(parents=True)
        Image.fromarray(g[f"src_left{i}"]).save(d / "left.png")
        Image.fromarray(g[f"src_right{i}"]).save(d / "right.png")
        S.write_pfm(str(d / "disp.pfm"), g[f"src_disp{i}"])
        lines.append(f"frames/{i}/left.png frames/{i}/right.png frames/{i}/disp.pfm")
    lst = tmp_path / "train.list"
    lst.write_text("\n".join(lines) + "\n")
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    ds = S.StereoDataset(str(tmp_path), str(lst), str(lst), mode="train", save_filename=True,
                         transform=T.Compose([T.RandomCrop(16, 32, validate=True), T.ToTensor(), T.Normalize(mean, std)]))
    assert len(ds) == 2 and ds.get_img_size() == (540, 960) and ds.get_scale_size() == (576, 960)
    for i in range(2):
        smp = ds[i]
        np.testing.assert_allclose(smp["img_left"].numpy(), g[f"crop_left{i}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(smp["img_right"].numpy(), g[f"crop_right{i}"], rtol=0, atol=1e-6)
        assert np.array_equal(smp["gt_disp"].numpy(), g[f"crop_disp{i}"])
        assert smp["left_name"] == str(g[f"name{i}"][0])
    ds2 = S.StereoDataset(str(tmp_path), str(lst), str(lst), mode="train", transform=T.Compose([T.RandomCrop(24, 40), T.ToTensor()]))
    smp = ds2[1]   # zero padding at the top / right (transforms.py:75-104; unpinned: np.lib.pad is gone from numpy 2)
    assert smp["img_left"].shape == (3, 24, 40) and smp["gt_disp"].shape == (24, 40)
    assert np.array_equal(smp["gt_disp"].numpy()[4:, :36], g["src_disp1"]) and float(smp["gt_disp"][:4].abs().sum()) == 0.0
    assert float(smp["img_left"][:, :, 36:].abs().sum()) == 0.0
    np.testing.assert_allclose(smp["img_left"].numpy()[:, 4:, :36], np.transpose(g["src_left1"], (2, 0, 1)) / 255.0, atol=1e-6)
    Image.fromarray(g["kitti_png"]).save(tmp_path / "kitti_disp.png")
    assert np.array_equal(T.read_disp(str(tmp_path / "kitti_disp.png")), g["kitti_disp"])
    # val mode: images resized to 576x960-style scale size (here: a small stand-in size), disparity untouched
    dsv = S.StereoDataset(str(tmp_path), str(lst), str(lst), mode="val")
    dsv.scale_size = (24, 36)
    v = dsv[0]
    assert v["img_left"].shape == (24, 36, 3) and v["gt_disp"].shape == (20, 36) and v["img_left"].dtype == np.float32
