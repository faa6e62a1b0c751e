"""PFM disparity files (SceneFlow ground truth), SURVEY 8(f) N4.  Same container format the reference reads
and writes in ``utils/file_io.py:34-99`` (header ``Pf``/``PF``, ``width height``, signed scale = endianness,
rows stored bottom-up)."""
from __future__ import annotations

import re
import sys

import numpy as np


def read_pfm(path):
    """-> (data [H,W] or [H,W,3] float32, top row first; scale)."""
    with open(path, "rb") as f:
        header = f.readline().rstrip().decode("ascii")
        if header not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("ascii"))
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().decode("ascii").rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (height, width, 3) if header == "PF" else (height, width)
    return np.flipud(data.reshape(shape)).astype(np.float32), abs(scale)


def write_pfm(path, image, scale=1):
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    flipped = np.flipud(image)
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    with open(path, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(b"%d %d\n" % (flipped.shape[1], flipped.shape[0]))
        f.write(b"%f\n" % (-scale if little else scale))
        flipped.tofile(f)


def read_disp(path, subset=False):
    """``read_disp`` for .pfm / .npy (``file_io.py:17-31``; the KITTI .png branch needs PIL and is not built)."""
    if path.endswith("pfm"):
        disp = np.ascontiguousarray(read_pfm(path)[0])
        return -disp if subset else disp
    if path.endswith("npy"):
        return np.load(path)
    raise Exception("Invalid disparity file format!")
