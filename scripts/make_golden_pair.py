#!/usr/bin/env python3
"""Decodes the rectified stereo pair the reference bundles (data/rectifiedImg/rectify{L,R}.jpg, 2560x1440)
into small grayscale PGM fixtures under tests/golden/ (640x360: 4x4 block mean of the gray image).

JPEG decoding (PIL/libjpeg here, OpenCV's decoder in the reference) is not guaranteed bit-identical, which is
why the decoded pixels, not the JPEGs, are the fixture.  Gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14, the
fixed-point formula of cv::cvtColor(COLOR_BGR2GRAY)."""
import os
import sys

import numpy as np
from PIL import Image

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/cameraCalibration/data/rectifiedImg"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

for side in "LR":
    rgb = np.asarray(Image.open(os.path.join(SRC, f"rectify{side}.jpg")).convert("RGB")).astype(np.int64)
    gray = (rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14
    h, w = gray.shape
    small = gray.reshape(h // 4, 4, w // 4, 4).sum(axis=(1, 3))
    small = ((small + 8) // 16).astype(np.uint8)
    path = os.path.join(DST, f"bundled_rectify{side}_640x360.pgm")
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (small.shape[1], small.shape[0]))
        f.write(small.tobytes())
    print(path, small.shape, int(small.mean()))
