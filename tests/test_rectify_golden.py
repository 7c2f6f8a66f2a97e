"""stereoRectify against the one known answer the reference ships (tests/golden/stereo_rectify_2560x1440.json,
extracted from cameraCalibration/unditortStereoCalib.xml by scripts/make_golden_rectify.py): both the CPU
oracle and the library's host routine must reproduce Rectify_R1/R2/P1/P2/Q.  Host-only, no GPU."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    g = json.load(open(os.path.join(HERE, "golden", "stereo_rectify_2560x1440.json")))
    return {k: (np.array(v["data"]).reshape(v["rows"], v["cols"]) if isinstance(v, dict) else v) for k, v in g.items()}


def _check(golden, R1, R2, P1, P2, Q):
    # rotations: double arithmetic end to end -> 1e-14; projections: the principal point comes from float32
    # corner projections, reproduced exactly (tolerance = 1e-9 px to allow for printing)
    assert np.abs(R1 - golden["Rectify_R1"]).max() < 1e-14
    assert np.abs(R2 - golden["Rectify_R2"]).max() < 1e-14
    assert np.abs(P1 - golden["Rectify_P1"]).max() < 1e-9
    assert np.abs(P2 - golden["Rectify_P2"]).max() < 1e-7      # P2[0,3] = -1.3e5
    assert np.abs(Q - golden["Rectify_Q"]).max() < 1e-9


def test_oracle_reproduces_reference_rectification(orc, golden):
    r = orc.stereo_rectify(golden["StereoCalib_K1"], golden["StereoCalib_D1"], golden["StereoCalib_K2"], golden["StereoCalib_D2"],
                           tuple(golden["ImgSize"]), golden["StereoCalib_R"], golden["StereoCalib_T"], alpha=-1)
    _check(golden, r["R1"], r["R2"], r["P1"], r["P2"], r["Q"])


def test_library_reproduces_reference_rectification(api, golden):
    R1, R2, P1, P2, Q, roi1, roi2 = api.stereoRectify(golden["StereoCalib_K1"], golden["StereoCalib_D1"], golden["StereoCalib_K2"],
                                                      golden["StereoCalib_D2"], tuple(golden["ImgSize"]), golden["StereoCalib_R"],
                                                      golden["StereoCalib_T"], alpha=-1)
    _check(golden, R1, R2, P1, P2, Q)
    assert 0 <= roi1[0] < 200 and roi1[2] > 2000 and roi2[3] > 1000


@pytest.mark.parametrize("alpha", [-1.0, 0.0, 0.5, 1.0])
def test_library_matches_oracle_for_every_alpha(api, orc, golden, alpha):
    args = (golden["StereoCalib_K1"], golden["StereoCalib_D1"], golden["StereoCalib_K2"], golden["StereoCalib_D2"],
            tuple(golden["ImgSize"]), golden["StereoCalib_R"], golden["StereoCalib_T"])
    R1, R2, P1, P2, Q, roi1, roi2 = api.stereoRectify(*args, alpha=alpha)
    r = orc.stereo_rectify(*args, alpha=alpha)
    for a, b in ((R1, r["R1"]), (R2, r["R2"])):
        assert np.abs(a - b).max() < 1e-14
    for a, b in ((P1, r["P1"]), (P2, r["P2"]), (Q, r["Q"])):
        assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())
    assert roi1 == r["roi1"] and roi2 == r["roi2"]


def test_rectified_maps_send_principal_axis_to_principal_point(api, golden):
    # geometry check independent of any stored answer: the new principal point maps back onto the optical
    # ray R1^T * (0,0,1) of the left camera
    R1, R2, P1, P2, Q, _, _ = api.stereoRectify(golden["StereoCalib_K1"], golden["StereoCalib_D1"], golden["StereoCalib_K2"],
                                                golden["StereoCalib_D2"], tuple(golden["ImgSize"]), golden["StereoCalib_R"],
                                                golden["StereoCalib_T"], alpha=-1)
    assert np.allclose(R2.T @ R1, golden["StereoCalib_R"], atol=1e-13)       # relative rotation is preserved
    t = R2 @ golden["StereoCalib_T"].ravel()
    assert abs(t[1]) < 1e-10 and abs(t[2]) < 1e-10                            # baseline along x after rectification
    assert P2[0, 3] == pytest.approx(t[0] * P2[0, 0], rel=1e-12)
