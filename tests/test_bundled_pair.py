"""Config "bundled pair": the rectified stereo pair the reference ships (tests/golden/bundled_rectify{L,R}_640x360.pgm,
made by scripts/make_golden_pair.py), rectified with the calibration of tests/golden/stereo_rectify_2560x1440.json
scaled to the fixture size, then run through the disparity routines.

CPU part: plumbing through the oracle only.  GPU part: every stage of the library equals the oracle on real image
content (bit-exact: all integer stages)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = (int(v) for v in f.readline().split())
        assert int(f.readline()) == 255
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w).copy()


@pytest.fixture(scope="module")
def pair():
    return (read_pgm(os.path.join(HERE, "golden", "bundled_rectifyL_640x360.pgm")),
            read_pgm(os.path.join(HERE, "golden", "bundled_rectifyR_640x360.pgm")))


@pytest.fixture(scope="module")
def calib():
    g = json.load(open(os.path.join(HERE, "golden", "stereo_rectify_2560x1440.json")))
    m = {k: np.array(v["data"]).reshape(v["rows"], v["cols"]) for k, v in g.items() if isinstance(v, dict)}
    s = np.diag([0.25, 0.25, 1.0])                 # 2560x1440 -> 640x360
    m["K1s"], m["K2s"] = s @ m["StereoCalib_K1"], s @ m["StereoCalib_K2"]
    return m


def _rectify(mod_rectify, mod_maps, mod_remap, calib, img, left):
    args = (calib["K1s"], calib["StereoCalib_D1"], calib["K2s"], calib["StereoCalib_D2"], (640, 360), calib["StereoCalib_R"],
            calib["StereoCalib_T"])
    r = mod_rectify(*args)
    if isinstance(r, dict):
        R1, R2, P1, P2 = r["R1"], r["R2"], r["P1"], r["P2"]
    else:
        R1, R2, P1, P2 = r[:4]
    K, D, Rk, Pk = (calib["K1s"], calib["StereoCalib_D1"], R1, P1) if left else (calib["K2s"], calib["StereoCalib_D2"], R2, P2)
    m1, m2 = mod_maps(K, D, Rk, Pk, (640, 360))
    return mod_remap(img, m1, m2), (m1, m2)


def test_plumbing_through_the_cpu_oracle(orc, pair, calib):
    L, R = pair
    rl, _ = _rectify(lambda *a: orc.stereo_rectify(*a, alpha=0.0), orc.init_undistort_rectify_map, orc.remap, calib, L, True)
    rr, _ = _rectify(lambda *a: orc.stereo_rectify(*a, alpha=0.0), orc.init_undistort_rectify_map, orc.remap, calib, R, False)
    assert rl.shape == rr.shape == (360, 640) and rl.std() > 10 and rr.std() > 10
    # the bundled pair is already rectified: run the reference's census routine on it directly
    d = orc.census_stereo(L, R, 7, 0, 63, True, orc.BASIC_CENSUS, orc.AGG_BOX)
    assert d.dtype == np.float64 and d.shape == L.shape
    assert 0 <= d.min() and d.max() <= 63
    d8 = orc.sad_box(L, R, 7, 0, 63, True)
    assert d8.dtype == np.uint8
    # left-disparity of a rectified pair is mostly positive
    assert np.median(d[:, 64:]) > 0


@pytest.mark.gpu
def test_library_equals_oracle_on_the_bundled_pair(api, orc, pair, calib):
    L, R = pair
    # rectification: maps and remap
    for img, left in ((L, True), (R, False)):
        got, gm = _rectify(lambda *a: api.stereoRectify(*a, alpha=0.0), api.initUndistortRectifyMap, api.remap, calib, img, left)
        exp, em = _rectify(lambda *a: orc.stereo_rectify(*a, alpha=0.0), orc.init_undistort_rectify_map, orc.remap, calib, img, left)
        assert np.array_equal(gm[0], em[0]) and np.array_equal(gm[1], em[1])
        assert np.array_equal(got, exp)
    # disparity routines
    p = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=7, minDisparity=0, maxDisparity=63, isDispLeft=True)
    assert np.array_equal(api.censusStereo(p, api.BASIC_CENSUS, api.AGG_BOX), orc.census_stereo(L, R, 7, 0, 63, True, orc.BASIC_CENSUS, orc.AGG_BOX))
    assert np.array_equal(api.computeSAD_BFOpti(p), orc.sad_box(L, R, 7, 0, 63, True))
    assert np.array_equal(api.computeSAD_inteOpti(p), orc.sad_inte(L, R, 7, 0, 63, True, orc.SAT_F32))
    sp = api.sgbm_params(0, 64, 8, 32, 10, 1, 8, 1)
    assert np.array_equal(api.sgbm_compute(L, R, sp), orc.sgm(L, R, 0, 64, 8, 32, 10, 1, 8, 1))
