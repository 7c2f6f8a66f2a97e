#!/usr/bin/env python3
"""Extracts the rectification known answer the reference ships into tests/golden/ (data only).

Source: /root/reference/cameraCalibration/unditortStereoCalib.xml -- an OpenCV FileStorage dump that
holds the inputs (StereoCalib_K1/D1/K2/D2/R/T, ImgSize) and the outputs of cv::stereoRectify for them
(Rectify_R1/R2/P1/P2/Q).  Only the numbers are copied, as JSON."""
import json
import os
import sys
import xml.etree.ElementTree as ET

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/cameraCalibration/unditortStereoCalib.xml"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "stereo_rectify_2560x1440.json")

root = ET.parse(SRC).getroot()
out = {"source": "cameraCalibration/unditortStereoCalib.xml", "ImgSize": [int(v) for v in root.find("ImgSize").text.split()]}
for node in root:
    if node.get("type_id") == "opencv-matrix":
        rows, cols = int(node.find("rows").text), int(node.find("cols").text)
        if rows * cols == 0:
            continue
        vals = [float(v) for v in node.find("data").text.split()]
        assert len(vals) == rows * cols
        out[node.tag] = {"rows": rows, "cols": cols, "data": vals}
json.dump(out, open(DST, "w"), indent=1)
print(DST, sorted(out))
