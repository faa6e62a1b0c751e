"""Dataset + sample transforms for real-data evaluation (the dataset half).

Host-side mirror of ``dataloader/SceneflowLoader.py:13-143`` (``StereoDataset``: same constructor arguments, same sample
dictionary keys), ``utils/file_io.py:11-31,104-106`` (image / disparity readers) and the geometric transforms of
``dataloader/transforms.py:10-153`` (``Compose``, ``ToTensor``, ``Normalize``, ``RandomCrop``, ``RandomVerticalFlip``), so
that the metrics harness (``metrics.py``) can consume a mounted SceneFlow / KITTI tree.  Plain numpy / PIL on the host: data
IO is not on the hot path.

Not restated: the colour augmentations (``transforms.py:155-262`` are torchvision calls) and ``transform.resize`` of the
validation branch (``SceneflowLoader.py:124-130`` uses scikit-image, absent here): the val/test resize to 576x960 is done with
a half-pixel-centre bilinear resample -- the interior matches scikit-image's order-1 resize, the border rows differ by its
reflect padding; that one step is parity-unpinned and said so in DESIGN.md.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .pfm import read_pfm


def read_text_lines(filepath):
    with open(filepath, "r") as f:
        return [ln.rstrip() for ln in f.readlines()]


def read_img(filename):
    """RGB float32 [H,W,3] in 0..255 (file_io.py:11-14)."""
    from PIL import Image
    return np.array(Image.open(filename).convert("RGB")).astype(np.float32)


def read_disp(filename, subset=False):
    """SceneFlow ``.pfm`` (negated for the 'subset' release), KITTI 16-bit ``.png`` / 256, or ``.npy`` (file_io.py:17-31)."""
    if filename.endswith("pfm"):
        disp = np.ascontiguousarray(read_pfm(filename)[0])
        return -disp if subset else disp
    if filename.endswith("png"):
        from PIL import Image
        return np.array(Image.open(filename)).astype(np.float32) / 256.0
    if filename.endswith("npy"):
        return np.load(filename)
    raise Exception("Invalid disparity file format!")


def _resize_bilinear_hwc(img, size):
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).permute(2, 0, 1)[None]
    out = torch.nn.functional.interpolate(t, size=tuple(size), mode="bilinear", align_corners=False)
    return out[0].permute(1, 2, 0).contiguous().numpy().astype(np.float32)


class StereoDataset(Dataset):
    """``StereoDataset(data_dir, train_datalist, test_datalist, dataset_name='SceneFlow', mode='train',
    save_filename=False, load_pseudo_gt=False, transform=None)`` (SceneflowLoader.py:13-143).

    Samples: ``img_left`` / ``img_right`` [H,W,3] float32, ``gt_disp`` [H,W] (when the list has a third column),
    ``pseudo_disp`` (KITTI, optional), ``left_name`` (``save_filename``).  In 'val' / 'test' mode the images are resized
    to 576x960 (:124-130; the prediction is rescaled back by ``metrics.scale_disp``)."""

    _KITTI = {
        "KITTI2012": {"train": "filenames/KITTI_2012_train.txt", "train_all": "filenames/KITTI_2012_train_all.txt",
                      "val": "filenames/KITTI_2012_val.txt", "test": "filenames/KITTI_2012_test.txt"},
        "KITTI2015": {"train": "filenames/KITTI_2015_train.txt", "train_all": "filenames/KITTI_2015_train_all.txt",
                      "val": "filenames/KITTI_2015_val.txt", "test": "filenames/KITTI_2015_test.txt"},
        "KITTI_mix": {"train": "filenames/KITTI_mix.txt", "test": "filenames/KITTI_2015_test.txt"},
    }

    def __init__(self, data_dir, train_datalist, test_datalist, dataset_name="SceneFlow", mode="train",
                 save_filename=False, load_pseudo_gt=False, transform=None):
        super().__init__()
        self.data_dir, self.dataset_name, self.mode = data_dir, dataset_name, mode
        self.save_filename, self.transform = save_filename, transform
        self.train_datalist, self.test_datalist = train_datalist, test_datalist
        self.img_size, self.scale_size = (540, 960), (576, 960)
        lists = dict(self._KITTI, SceneFlow={"train": train_datalist, "val": test_datalist, "test": test_datalist})
        assert dataset_name in lists
        self.samples = []
        for line in read_text_lines(lists[dataset_name][mode]):
            splits = line.split()
            if not splits:
                continue
            left_img, right_img = splits[:2]
            gt_disp = None if len(splits) == 2 else splits[2]
            sample = {}
            if save_filename:
                sample["left_name"] = left_img.split("/", 1)[1]
            sample["left"] = os.path.join(data_dir, left_img)
            sample["right"] = os.path.join(data_dir, right_img)
            sample["disp"] = os.path.join(data_dir, gt_disp) if gt_disp is not None else None
            sample["pseudo_disp"] = None
            if load_pseudo_gt and sample["disp"] is not None:
                if "disp_occ_0" in sample["disp"]:
                    sample["pseudo_disp"] = sample["disp"].replace("disp_occ_0", "disp_occ_0_pseudo_gt")
                elif "disp_occ" in sample["disp"]:
                    sample["pseudo_disp"] = sample["disp"].replace("disp_occ", "disp_occ_pseudo_gt")
                else:
                    raise NotImplementedError
            self.samples.append(sample)

    def __getitem__(self, index):
        path = self.samples[index]
        sample = {}
        if self.save_filename:
            sample["left_name"] = path["left_name"]
        sample["img_left"] = read_img(path["left"])
        sample["img_right"] = read_img(path["right"])
        subset = "subset" in self.dataset_name
        if path["disp"] is not None:
            sample["gt_disp"] = read_disp(path["disp"], subset=subset)
        if path["pseudo_disp"] is not None:
            sample["pseudo_disp"] = read_disp(path["pseudo_disp"], subset=subset)
        if self.mode in ("test", "val"):
            sample["img_left"] = _resize_bilinear_hwc(sample["img_left"], self.scale_size)
            sample["img_right"] = _resize_bilinear_hwc(sample["img_right"], self.scale_size)
        if self.transform is not None:
            sample = self.transform(sample)
        return sample

    def __len__(self):
        return len(self.samples)

    def get_img_size(self):
        return self.img_size

    def get_scale_size(self):
        return self.scale_size


# ------------------------------------------------------------------ transforms (dataloader/transforms.py:10-153)
class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample


class ToTensor:
    """numpy -> torch: images [H,W,3] -> [3,H,W] / 255; disparities as they are (:19-43)."""

    def __call__(self, sample):
        for key in ("img_left", "img_right"):
            sample[key] = torch.from_numpy(np.ascontiguousarray(np.transpose(sample[key], (2, 0, 1)))) / 255.0
        for key in ("gt_disp", "pseudo_disp"):
            if key in sample:
                sample[key] = torch.from_numpy(np.ascontiguousarray(sample[key]))
        if "gt_normal" in sample:
            sample["gt_normal"] = torch.from_numpy(np.ascontiguousarray(np.transpose(sample["gt_normal"], (2, 0, 1))))
        return sample


class Normalize:
    """Per-channel (x - mean) / std on the image tensors (:46-62)."""

    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, sample):
        for key in ("img_left", "img_right"):
            for t, m, s in zip(sample[key], self.mean, self.std):
                t.sub_(m).div_(s)
        return sample


class RandomCrop:
    """Random (training) or centre (``validate``) crop; smaller images are zero-padded at the top / right (:65-132)."""

    _KEYS3 = ("img_left", "img_right", "gt_normal")
    _KEYS2 = ("gt_disp", "pseudo_disp")

    def __init__(self, img_height, img_width, validate=False):
        self.img_height, self.img_width, self.validate = img_height, img_width, validate

    def __call__(self, sample):
        ori_height, ori_width = sample["img_left"].shape[:2]
        if self.img_height > ori_height or self.img_width > ori_width:
            top_pad, right_pad = self.img_height - ori_height, self.img_width - ori_width
            assert top_pad >= 0 and right_pad >= 0
            for key in self._KEYS3:
                if key in sample:
                    sample[key] = np.pad(sample[key], ((top_pad, 0), (0, right_pad), (0, 0)), mode="constant", constant_values=0)
            for key in self._KEYS2:
                if key in sample:
                    sample[key] = np.pad(sample[key], ((top_pad, 0), (0, right_pad)), mode="constant", constant_values=0)
            return sample
        if not self.validate:
            self.offset_x = np.random.randint(ori_width - self.img_width + 1)
            self.offset_y = np.random.randint(0, ori_height - self.img_height + 1)
        else:
            self.offset_x = (ori_width - self.img_width) // 2
            self.offset_y = (ori_height - self.img_height) // 2
        for key in self._KEYS3 + self._KEYS2:
            if key in sample:
                sample[key] = self.crop_img(sample[key])
        return sample

    def crop_img(self, img):
        return img[self.offset_y:self.offset_y + self.img_height, self.offset_x:self.offset_x + self.img_width]


class RandomVerticalFlip:
    """Flip all maps upside down with probability 0.5 (:135-151)."""

    def __call__(self, sample):
        if np.random.random() < 0.5:
            for key in ("img_left", "img_right", "gt_disp", "gt_normal", "pseudo_disp"):
                if key in sample:
                    sample[key] = np.copy(np.flipud(sample[key]))
        return sample
