"""The C-ABI library loads and exports every symbol include/mdisp.h declares (no GPU needed)."""
import ctypes
import os
import re

from cameracalibration_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mdisp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdisp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"libmdisp.so does not export {name}"
    assert set(declared) == set(_lib.EXPORTS)


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.mdisp_version() >= 100
    assert isinstance(lib.mdisp_last_error(), bytes)


def test_no_device_fails_loudly_without_fallback():
    lib = _lib.load()
    if lib.mdisp_device_count() > 0:
        return
    h = ctypes.c_void_p()
    rc = lib.mdisp_create(ctypes.byref(h), -1, None)
    assert rc == _lib.E_NODEVICE
    assert b"no CPU fallback" in lib.mdisp_last_error()


def test_multi_device_entry_points_fail_loudly_without_a_device():
    import numpy as np
    from cameracalibration_amd import api
    from cameracalibration_amd._lib import MdispError
    lib = _lib.load()
    if lib.mdisp_device_count() > 0:
        return
    img = np.zeros((32, 48), np.uint8)
    par = api.sgbm_params(0, 16, 8, 32, 10, 1, 8, 1)
    for call in (lambda: api.sgbm_compute_batch_multi([0], [img], [img], par), lambda: api.sgbm_compute_rows_multi([0], img, img, par)):
        try:
            call()
            raise AssertionError("expected MDISP_E_NODEVICE")
        except MdispError as e:
            assert e.code == _lib.E_NODEVICE and "no CPU fallback" in str(e)


def test_rectify_maps_and_synth_are_host_only():
    import numpy as np
    from cameracalibration_amd import api
    K = np.array([[500.0, 0, 31.5], [0, 500.0, 23.5], [0, 0, 1]])
    m1, m2 = api.initUndistortRectifyMap(K, np.zeros(5), np.eye(3), K, (64, 48))
    xs, ys = np.meshgrid(np.arange(64), np.arange(48))
    assert np.array_equal(m1[..., 0], xs) and np.array_equal(m1[..., 1], ys)
    assert not m2.any()
    left, right, gt = api.synth_pair(0x5EED0001, 48, 64, 16)
    assert left.shape == right.shape == gt.shape == (48, 64)
    assert right.max() == 255 and right.min() == 0 and gt.max() <= 15
